"""Multi-process (gloo, world_size 2) tests of the N > 1 path: series sharding with no data-path
collective, and the all-gather + ordered merge of the aggregate partials. The per-rank compute is
the CPU oracle here (no GPU in this container); on the GPU box the same code runs over RCCL."""

import os
import socket

import numpy as np
import pytest

import datagen
import oracle_lib as ora
import modelardb_rs_amd as mdb
from modelardb_rs_amd import sharding

ALL = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
N_SERIES, N_POINTS = 6, 20_000


def _series_batch(series):
    eb = mdb.error_bound("relative", 1.0)
    timestamps, values = datagen.sine_series(series, N_POINTS)
    return ora.try_compress_univariate_time_series(timestamps, values, eb)


def test_series_ranges_partition_the_series():
    for n_series in (1, 7, 8, 1000, 100_000):
        for world in (1, 2, 3, 8):
            covered = []
            for rank in range(world):
                first, last = sharding.series_range(n_series, rank, world)
                covered += list(range(first, last)) if n_series <= 1000 else [(first, last)]
                if n_series <= 1000:
                    assert all(sharding.owner_of_series(s, n_series, world) == rank
                               for s in range(first, last))
            if n_series <= 1000:
                assert covered == list(range(n_series))
            else:
                assert covered[0][0] == 0 and covered[-1][1] == n_series
                assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def test_pack_unpack_preserves_bits():
    state = mdb._abi.AggStateC(-1.25e300, -(1 << 62), float("-inf"), 3.4028234663852886e38)
    back = sharding.unpack_state(sharding.pack_state(state))
    assert (back.sum, back.count, back.min, back.max) == (state.sum, state.count, state.min, state.max)


def _worker(rank, world, port, queue):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = sharding.series_range(N_SERIES, rank, world)
    partial = mdb._abi.AggStateC.fresh()
    grid_points = 0
    for series in range(first, last):
        batch = _series_batch(series)
        partial = ora.agg_batch(batch, ALL, partial)
        grid_points += len(ora.grid_batch(batch)[0])
    merged = sharding.all_reduce_state(partial, dist)
    queue.put((rank, merged.sum, merged.count, merged.min, merged.max, grid_points))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_aggregate_merge_matches_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    context = mp.get_context("spawn")
    queue = context.Queue()
    world = 2
    processes = [context.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in processes:
        p.start()
    results = sorted(queue.get(timeout=120) for _ in range(world))
    for p in processes:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = mdb._abi.AggStateC.fresh()
    for series in range(N_SERIES):
        expected = ora.agg_batch(_series_batch(series), ALL, expected)
    for rank, total, count, mn, mx, _ in results:
        assert count == expected.count == N_SERIES * N_POINTS
        assert (mn, mx) == (expected.min, expected.max)
        assert abs(total - expected.sum) <= 1e-12 * abs(expected.sum)
    assert results[0][1:5] == results[1][1:5]          # every rank holds the same merged state
    assert sum(r[5] for r in results) == N_SERIES * N_POINTS   # grid shards cover every point once


def test_merge_order_is_rank_order():
    states = [mdb._abi.AggStateC(0.1, 1, 5.0, 5.0), mdb._abi.AggStateC(0.2, 2, -1.0, 9.0),
              mdb._abi.AggStateC(0.3, 3, 2.0, 2.0)]
    merged = sharding.merge_states(states)
    assert merged.sum == (0.1 + 0.2) + 0.3 and merged.count == 6
    assert (merged.min, merged.max) == (-1.0, 9.0)
