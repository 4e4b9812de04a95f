#!/usr/bin/env python3
"""bench.py - headline benchmark of the ModelarDB hot path on MI355X.

One "step" = one pass of grid() (segment -> data point reconstruction, the GridExec hot loop) over
the whole synthetic batch of BASELINE.json configs[1]: 1 000 series x 10 000 000 points of
sine + noise compressed with a 1 % relative error bound, per GPU (weak scaling: every rank holds its
own 1k x 10M shard; series shard embarrassingly, no data-path collective). Segments are resident in
HBM before the timed region and the reconstructed (timestamp, value) columns are written to HBM.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the dominant
kernel (k_grid_tiles: algorithmic bytes / HIP-event time measured on the launch stream) and
`cpu_baseline` (the CPU oracle, a port of the reference's per-row GridStream loop, timed on the
host cores on a bounded sample of the same segments).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--series S] [--points P]
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO_ROOT = os.path.dirname(os.path.abspath(__file__))
for _path in (REPO_ROOT, os.path.join(REPO_ROOT, "tests")):
    if _path not in sys.path:
        sys.path.insert(0, _path)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured)
CHUNK_POINTS = 65536    # ingest buffer of the reference server (storage/mod.rs:58)
SEED = 0x4D44425F52454631
INTERVAL_US = 1000


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=5)
    parser.add_argument("--warmup", type=int, default=2)
    parser.add_argument("--series", type=int, default=1000)
    parser.add_argument("--points", type=int, default=10_000_000)
    parser.add_argument("--error-bound", type=float, default=1.0, help="relative bound in percent")
    parser.add_argument("--cpu-sample-series", type=int, default=48)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--range-middle", type=float, default=0.0,
                        help="BASELINE config 5: the timed step is a point-range GridExec query over this "
                             "fraction of the time axis (centred), e.g. 0.5; 0 = the whole series (config 2)")
    parser.add_argument("--settle-seconds", type=float, default=0.0,
                        help="run the step untimed for this long before the warmup (lets clocks settle)")
    parser.add_argument("--fit-group-points", type=int, default=13_000_000_000,
                        help="at most this many raw points (4 B each) are resident per fit launch")
    return parser.parse_args()


def init_distributed(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # Launched by torch.distributed.run (any world size, so the 1-GPU box exercises the RCCL path too).
    if world > 1 or os.environ.get("TORCHELASTIC_RUN_ID") is not None:
        import torch
        import torch.distributed as dist_module
        torch.cuda.set_device(local_rank)
        dist_module.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_module
    return rank, local_rank, world, dist


def pmc_traffic(args, points_per_launch, segments_per_launch):
    """HBM bytes per launch of k_grid_tiles from the committed rocprofv3 PMC passes (bench.py cannot
    collect counters itself): WRITE_SIZE and FETCH_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950. Only reported for the workload the passes ran on."""
    path = os.path.join(REPO_ROOT, "profiles", "pmc_grid_tiles.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        pmc = json.load(f)
    if (pmc.get("series"), pmc.get("points")) != (args.series, args.points):
        return None
    return (pmc["write_bytes_per_point"] * points_per_launch
            + pmc["fetch_bytes_per_segment_corrected"] * segments_per_launch)


def eb_for(args, mdb):
    return mdb.error_bound("relative", args.error_bound)


def barrier_and_sync(context, dist):
    """Both sides of the timed region: this rank's launch stream drained, every rank arrived, and
    (the barrier is a collective on torch's stream) torch's streams drained too."""
    context.sync()
    if dist is not None:
        import torch
        dist.barrier()
        torch.cuda.synchronize()
    context.sync()


def fit_on_gpu(context, mdb, args, rank):
    """Generate this rank's series on the device and compress them with the HIP fitter, in groups of
    series so that raw values never need more than a few GB of HBM at once."""
    import ctypes as C
    eb = mdb.error_bound("relative", args.error_bound)
    chunks_per_series = (args.points + CHUNK_POINTS - 1) // CHUNK_POINTS
    # All series of the rank in one launch when they fit (one lane per chunk: occupancy = chunks).
    group = max(1, min(args.series, args.fit_group_points // max(args.points, 1)))
    parts, fit_seconds, fit_points = [], 0.0, 0
    first_series_of_rank = rank * args.series
    for first in range(0, args.series, group):
        n_series = min(group, args.series - first)
        total = n_series * args.points
        values = context.dev_alloc(4 * total)
        context.synth_values_dev(values, first_series_of_rank + first, n_series, args.points, SEED)
        offsets = np.zeros(n_series * chunks_per_series + 1, dtype=np.uint64)
        first_index = np.zeros(n_series * chunks_per_series, dtype=np.uint64)
        k = 0
        for s in range(n_series):
            for c in range(chunks_per_series):
                start = c * CHUNK_POINTS
                offsets[k] = s * args.points + start
                first_index[k] = start
                k += 1
        offsets[k] = total
        offsets_dev = context.upload_array(offsets)
        first_index_dev = context.upload_array(first_index)
        # The first call grows the context's scratch (tens of GB of hipMalloc); time the second.
        context.compress_chunks_dev(0, values, offsets_dev, k, eb, 0, INTERVAL_US, first_index_dev).free()
        context.sync()
        t0 = time.perf_counter()
        parts.append(context.compress_chunks_dev(0, values, offsets_dev, k, eb, 0, INTERVAL_US,
                                                 first_index_dev))
        context.sync()
        fit_seconds += time.perf_counter() - t0
        fit_points += total
        for pointer in (values, offsets_dev, first_index_dev):
            context.dev_free(pointer)
    return parts, fit_seconds, fit_points


def main():
    args = parse_args()
    rank, local_rank, world, dist = init_distributed(args.gpus)
    import modelardb_rs_amd as mdb

    context = mdb.Context(local_rank)
    info = context.device_info()

    # ---- build the workload: fit on the GPU, keep the segments in HBM ---------------------------
    parts, fit_seconds, fit_points = fit_on_gpu(context, mdb, args, rank)
    n_segments = sum(len(p) for p in parts)

    total_points = 0
    for part in parts:
        total_points += context.grid_count_dev(part)
    assert total_points == args.series * args.points, (total_points, args.series * args.points)
    largest = max(context.grid_count_dev(p) for p in parts)
    out_ts = context.dev_alloc(8 * total_points)
    out_val = context.dev_alloc(4 * total_points)

    ranged = 0.0 < args.range_middle < 1.0
    step_lo = int(args.points * (0.5 - args.range_middle / 2)) * INTERVAL_US
    step_hi = int(args.points * (0.5 + args.range_middle / 2)) * INTERVAL_US

    def step():
        at = 0
        metrics_total = None
        for part in parts:
            if ranged:
                n, metrics = context.grid_batch_range_dev(part, step_lo, step_hi, out_ts + 8 * at,
                                                          out_val + 4 * at, total_points - at)
            else:
                n, metrics = context.grid_batch_dev(part, out_ts + 8 * at, out_val + 4 * at,
                                                    total_points - at)
            at += n
            if metrics_total is None:
                metrics_total = dict(metrics)
            else:
                for key, value in metrics.items():
                    metrics_total[key] += value
        return at, metrics_total

    settle_until = time.perf_counter() + args.settle_seconds
    while time.perf_counter() < settle_until:
        step()
        context.sync()
    for _ in range(args.warmup):
        step()
    barrier_and_sync(context, dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        produced, metrics = step()
    barrier_and_sync(context, dist)
    elapsed = time.perf_counter() - t0
    assert ranged or produced == total_points
    points_per_step = produced

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel: HIP events on the launch stream ------------------------
    context.profile_enable(True)
    context.profile_reset()
    for _ in range(args.steps):
        step()
    profile = context.profile()
    context.profile_enable(False)
    launches, total_ms = profile.get("k_grid_tiles", (0, 0.0))
    kernel_ms = total_ms / max(launches, 1)
    # Algorithmic bytes of one launch (SURVEY 8(d) / BASELINE.md 3): 73 B per segment read + payloads
    # larger than 12 B (out of line) + 12 B per reconstructed point written. The tile kernel itself
    # reads a 48 B descriptor + 8 B offset per segment instead of the raw 73 B (the prepass did that),
    # so pricing it at 73 B/segment + 12 B/point is the figure the contract names.
    points_per_launch = points_per_step / len(parts)
    segments_per_launch = n_segments / len(parts)
    algorithmic_bytes = 73.0 * segments_per_launch + 12.0 * points_per_launch
    achieved_gbps = algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    # ---- secondary measurements on the same resident segments (not the headline) ----------------
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    t_lo, t_hi = (args.points // 4) * INTERVAL_US, (3 * args.points // 4) * INTERVAL_US
    for part in parts:  # warm
        context.agg_batch_dev(part, mask)
    context.profile_enable(True)
    context.profile_reset()
    barrier_and_sync(context, dist)
    t0 = time.perf_counter()
    state = None
    for part in parts:
        state = context.agg_batch_dev(part, mask, state)
    context.sync()
    agg_seconds = time.perf_counter() - t0
    t0 = time.perf_counter()
    range_state = None
    for part in parts:
        range_state = context.agg_batch_range_dev(part, t_lo, t_hi, mask, range_state)
    context.sync()
    range_seconds = time.perf_counter() - t0
    agg_profile = context.profile()
    context.profile_enable(False)
    if dist is not None:
        # The one exchange step of the path: merge the per-GPU aggregate partials over RCCL/xGMI.
        from modelardb_rs_amd import sharding
        t0 = time.perf_counter()
        state = sharding.all_reduce_state(state, dist, device=f"cuda:{local_rank}")
        range_state = sharding.all_reduce_state(range_state, dist, device=f"cuda:{local_rank}")
        reduce_seconds = time.perf_counter() - t0
    else:
        reduce_seconds = 0.0
    assert state.count == world * total_points, (state.count, world * total_points)
    aggregates = {
        "segments_per_s": n_segments / agg_seconds,
        "seconds": agg_seconds,
        "kernel_ms": agg_profile.get("k_agg_segments", (1, 0.0))[1] / max(agg_profile.get("k_agg_segments", (1, 0.0))[0], 1),
        "result": {"count": state.count, "min": state.min, "max": state.max, "sum": state.sum,
                   "avg": state.sum / max(state.count, 1)},
        "range": {"t_lo": t_lo, "t_hi": t_hi, "seconds": range_seconds,
                  "segments_per_s": n_segments / range_seconds,
                  "kernel_ms": agg_profile.get("k_agg_range", (1, 0.0))[1] / max(agg_profile.get("k_agg_range", (1, 0.0))[0], 1),
                  "count": range_state.count, "min": range_state.min, "max": range_state.max, "sum": range_state.sum},
        "final_reduce_seconds": reduce_seconds,
        "note": "COUNT/MIN/MAX/SUM on the resident segments (BASELINE config 3: no grid); the range "
                "variant clips to the middle half of the time axis; with N > 1 the partials of all "
                "ranks are merged by one all-gather over RCCL",
    }

    # ---- CPU baseline: the oracle's per-row grid loop on a bounded sample -------------------------
    cpu_baseline = None
    fit_cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        import oracle_lib as ora
        sample = parts[0].download()
        chunks_per_series = (args.points + CHUNK_POINTS - 1) // CHUNK_POINTS
        n_sample = min(args.cpu_sample_series, args.series)
        # Whole leading series: segments are ordered by chunk, chunks by series.
        sample = sample.take(np.nonzero(sample.chunk_index < n_sample * chunks_per_series)[0])
        cores = os.cpu_count() or 1
        timing = {}
        ts_cpu = ora.grid_batch(sample, n_threads=cores, timing=timing)[0]
        cpu_seconds = timing["seconds"]
        single = sample.take(np.nonzero(sample.chunk_index < chunks_per_series)[0])
        ts_single = ora.grid_batch(single, n_threads=1, timing=timing)[0]
        cpu_baseline = {
            "value": len(ts_cpu) / cpu_seconds,
            "unit": "values/s",
            "cores": cores,
            "kind": "port",
            "sample": f"grid() of the first {n_sample} series ({len(ts_cpu)} points, {len(sample)} "
                      f"segments) of the same workload, segment ranges sharded over {cores} host "
                      f"threads, output buffers pre-touched",
            "single_thread_value": len(ts_single) / timing["seconds"],
        }
        # The fitter's CPU baseline: the oracle's greedy compression of a few of the same series.
        n_fit = min(4, args.series)
        raw = context.dev_alloc(4 * n_fit * args.points)
        context.synth_values_dev(raw, rank * args.series, n_fit, args.points, SEED)
        host_values = context.download_array(raw, n_fit * args.points, np.float32)
        context.dev_free(raw)
        host_ts = np.tile(np.arange(args.points, dtype=np.int64) * INTERVAL_US, n_fit)
        offsets = np.array([s * args.points + c for s in range(n_fit)
                            for c in range(0, args.points, CHUNK_POINTS)] + [n_fit * args.points],
                           dtype=np.uint64)
        t0 = time.perf_counter()
        fitted = ora.compress_chunks(host_ts, host_values, offsets, eb_for(args, mdb), n_threads=cores)
        fit_cpu_seconds = time.perf_counter() - t0
        fit_cpu = {"points_per_s": n_fit * args.points / fit_cpu_seconds,
                   "segments_per_s": len(fitted) / fit_cpu_seconds, "cores": cores, "kind": "port",
                   "sample": f"{n_fit} series x {args.points} points, chunks sharded over {cores} threads"}

    if rank == 0:
        value = world * points_per_step * args.steps / elapsed
        result = {
            "metric": "gridded values/sec",
            "value": value,
            "unit": "values/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.series} series x {args.points} points sine+noise, "
                            f"relative error bound {args.error_bound} %, grid() decode of the "
                            f"segments to (timestamp i64, value f32) columns in HBM, per GPU"
                            + (f"; point-range query over the middle {args.range_middle:g} of the time axis "
                               f"({points_per_step} points per step)" if ranged else ""),
                "series_per_gpu": args.series,
                "points_per_series": args.points,
                "segments_per_gpu": n_segments,
                "segment_mix": metrics,
                "parallelism": f"series-sharded x{world}, no data-path collective",
                "device": info["name"],
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_grid_tiles",
                "achieved": achieved_gbps,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved_gbps / HBM_PEAK_GBPS,
                "traffic": pmc_traffic(args, points_per_launch, segments_per_launch),
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": algorithmic_bytes,
                "other_kernels_ms": {name: ms / max(n, 1) for name, (n, ms) in profile.items()
                                     if name != "k_grid_tiles"},
            },
            "cpu_baseline": cpu_baseline,
            "aggregates": aggregates,
            "fit": {
                "cpu_baseline": fit_cpu,
                "points_per_s": fit_points / fit_seconds if fit_seconds > 0 else None,
                "segments_per_s": n_segments / fit_seconds if fit_seconds > 0 else None,
                "seconds": fit_seconds,
                "note": "PMC-Mean/Swing/MacaqueV fit of this rank's series on the GPU (setup, not "
                        "in the timed region), regular timestamps synthesised on the fly",
            },
        }
        print(json.dumps(result))

    for part in parts:
        part.free()
    context.dev_free(out_ts)
    context.dev_free(out_val)
    context.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
