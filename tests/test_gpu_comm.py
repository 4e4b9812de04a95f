"""The final aggregate merge behind the C ABI (mdb_comm_init / mdb_agg_all_reduce) on
one GPU: a communicator of world size 1 over RCCL. The fold rule itself (rank order, the
accumulators' update rules) is covered on CPU by tests/test_sharding_cpu.py through mdb_agg_merge;
world size 2 runs under `bench.py --gpus 2` on a multi-GPU node."""

import numpy as np
import pytest

import cases
import modelardb_rs_amd as mdb

pytestmark = pytest.mark.gpu

ALL = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM


def test_all_reduce_with_one_rank_returns_the_state_itself():
    context = mdb.Context(0)
    try:
        with pytest.raises(mdb.HipError, match="mdb_comm_init"):
            context.agg_all_reduce(mdb._abi.AggStateC.fresh())
        unique_id = mdb.comm_unique_id()
        assert len(unique_id) == 128
        context.comm_init(0, 1, unique_id)
        with pytest.raises(mdb.HipError, match="already"):
            context.comm_init(0, 1, unique_id)
        batch = cases.mixed_batch(cases.error_bounds()["rel5"], False, seed=7)[2]
        state = context.agg_batch(batch, ALL)
        for _ in range(3):
            merged, seen = context.agg_all_reduce(state)
            assert seen == 1
            assert (merged.count, merged.min, merged.max) == (state.count, state.min, state.max)
            assert np.float64(merged.sum).view(np.uint64) == np.float64(state.sum).view(np.uint64)
        # special values survive the wire bit for bit
        odd = mdb._abi.AggStateC(-0.0, -(1 << 62), float("-inf"), 3.4028234663852886e38)
        merged, _ = context.agg_all_reduce(odd)
        assert (merged.count, merged.min, merged.max) == (odd.count, odd.min, odd.max)
        context.comm_close()
        context.comm_close()  # idempotent
        with pytest.raises(mdb.HipError):
            context.agg_all_reduce(state)
    finally:
        context.close()


def test_invalid_ranks_are_errors():
    context = mdb.Context(0)
    try:
        unique_id = mdb.comm_unique_id()
        for rank, world in ((-1, 1), (1, 1), (0, 0)):
            with pytest.raises(mdb.HipError):
                context.comm_init(rank, world, unique_id)
    finally:
        context.close()
