#!/usr/bin/env python3
"""bench.py - headline benchmark of the ModelarDB hot path on MI355X.

One "step" = one pass of grid() (segment -> data point reconstruction, the GridExec hot loop) over
the whole synthetic batch of BASELINE.json configs[1]: 1 000 series x 10 000 000 points of
sine + noise compressed with a 1 % relative error bound, per GPU (weak scaling: every rank holds its
own 1k x 10M shard; series shard embarrassingly, no data-path collective). Segments are resident in
HBM before the timed region and the reconstructed (timestamp, value) columns are written to HBM.

`python bench.py --gpus N` starts the N ranks itself when it is not already one of them (a child
`python -m torch.distributed.run`, started before this process touches a GPU; its JSON line is
relayed and its exit code returned). With N = 1 the single rank still joins a 1-rank process group
and an RCCL communicator of the C ABI (mdb_comm_init), so the final aggregate merge - the one
exchange step of the path - runs over RCCL on every box.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the dominant
kernel (k_grid_tiles: algorithmic bytes / HIP-event time measured on the launch stream),
`cpu_baseline` (the CPU oracle, a port of the reference's per-row GridStream loop, timed on the host
cores on a bounded sample of the same segments) and `verified`: what the run compared with the
oracle AFTER the timed region (the fitted segments of the sample series byte for byte, the
reconstructed columns of the sample bit for bit).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--series S] [--points P]
"""

import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

REPO_ROOT = os.path.dirname(os.path.abspath(__file__))
for _path in (REPO_ROOT, os.path.join(REPO_ROOT, "tests")):
    if _path not in sys.path:
        sys.path.insert(0, _path)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured)
CHUNK_POINTS = 65536    # ingest buffer of the reference server (storage/mod.rs:58)
SEED = 0x4D44425F52454631
INTERVAL_US = 1000
GRID_KERNEL_SOURCES = ("mdb_grid.hip", "mdb_segment_dev.hpp", "mdb_common.hpp")
FIT_KERNEL_SOURCES = ("mdb_fit.hip", "mdb_segment_dev.hpp", "mdb_common.hpp")
SHADER_CLOCK_MHZ = 2400.0  # hipDeviceAttributeClockRate of the MI355X (s_memtime against the host clock: 2.37-2.40 GHz under load)


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=5)
    parser.add_argument("--warmup", type=int, default=2)
    parser.add_argument("--series", type=int, default=1000)
    parser.add_argument("--points", type=int, default=10_000_000)
    parser.add_argument("--error-bound", type=float, default=1.0, help="relative bound in percent")
    parser.add_argument("--cpu-sample-series", type=int, default=48)
    parser.add_argument("--fit-sample-series", type=int, default=16)
    parser.add_argument("--no-cpu-baseline", action="store_true",
                        help="skip the CPU legs AND the in-run verification against the oracle")
    parser.add_argument("--no-irregular", action="store_true",
                        help="skip the block of series with irregular timestamps")
    parser.add_argument("--no-host-path", action="store_true",
                        help="skip the end-to-end GridStream (PCIe-inclusive) measurement")
    parser.add_argument("--no-mixed-models", action="store_true",
                        help="skip the block of Constant / Linear / Random data (all three model types)")
    parser.add_argument("--mixed-points", type=int, default=1_000_000_000)
    parser.add_argument("--full-tail", action="store_true",
                        help="with --gpus > 1: rank 0 also runs the host path, irregular and mixed-model blocks "
                             "(by default a multi-GPU run is short: the other ranks wait for rank 0's tail)")
    parser.add_argument("--collective-timeout", type=float, default=900.0,
                        help="seconds a rank waits in a collective before the job is torn down")
    parser.add_argument("--range-middle", type=float, default=0.0,
                        help="BASELINE config 5: the timed step is a point-range GridExec query over this "
                             "fraction of the time axis (centred), e.g. 0.5; 0 = the whole series (config 2)")
    parser.add_argument("--settle-seconds", type=float, default=0.0,
                        help="run the step untimed for this long before the warmup (lets clocks settle)")
    parser.add_argument("--fit-group-points", type=int, default=13_000_000_000,
                        help="at most this many raw points (4 B each) are resident per fit launch")
    parser.add_argument("--timed", choices=("grid", "fit"), default="grid",
                        help="what one timed step is: grid() of the rank's resident segments (BASELINE configs 2 / 5, "
                             "the default) or the PMC-Mean / Swing / MacaqueV fit of the rank's resident values "
                             "(BASELINE config 4: one mdb_compress_chunks_dev per step)")
    parser.add_argument("--detail-file", default=None,
                        help="where everything that was measured goes as one JSON object (default: bench_detail.json "
                             "beside bench.py); stdout carries the compact line only")
    parser.add_argument("--fit-repetitions", type=int, default=5,
                        help="timed repetitions of every fit that is not the timed step (median, min, max reported)")
    return parser.parse_args()


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks_if_needed(args):
    """`bench.py --gpus N` with N > 1 outside a launcher: start the N ranks as a CHILD process - this
    process has made no GPU call yet and never will - relay its output and leave with its code."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    environment = dict(os.environ)
    environment.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (a retry is a fresh child: a process that has touched the GPU is never re-executed)
    child = subprocess.run(command, env=environment, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(child.stdout)
    sys.stdout.flush()
    sys.exit(child.returncode)


def init_distributed(args, backend="nccl", timeout=None):
    """Every run is a torch.distributed job over RCCL, the single-GPU one included (a process group
    of one rank), so the collective path is exercised wherever the bench runs. (backend "gloo": the
    orchestration test of tests/test_bench_cpu.py, two ranks on CPU over a canned workload.)"""
    launched = "WORLD_SIZE" in os.environ
    if not launched:
        os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port())})
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    options = {"timeout": timeout} if timeout is not None else {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), **options)
    else:
        dist.init_process_group(backend=backend, **options)
    return rank, local_rank, world, dist


def source_hash(names):
    digest = hashlib.sha256()
    for name in names:
        with open(os.path.join(REPO_ROOT, "modelardb-rs_amd", "csrc", name), "rb") as f:
            digest.update(f.read())
    return digest.hexdigest()[:16]


def pmc_traffic(args, points_per_launch, segments_per_launch):
    """HBM bytes per launch of k_grid_tiles from the committed rocprofv3 PMC passes (bench.py cannot
    collect counters itself): WRITE_SIZE and FETCH_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950. Only reported for the workload the passes ran on AND
    while the kernel sources are the ones the passes measured (scripts/gpu_profile.sh stores their
    hash next to the counters): otherwise null and the reason."""
    path = os.path.join(REPO_ROOT, "profiles", "pmc_grid_tiles.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_grid_tiles.json is missing"
    with open(path) as f:
        pmc = json.load(f)
    if (pmc.get("series"), pmc.get("points")) != (args.series, args.points):
        return None, "the PMC passes ran on another workload"
    if pmc.get("source_hash") != source_hash(GRID_KERNEL_SOURCES):
        return None, ("the grid kernel sources changed since the PMC passes "
                      f"(measured {pmc.get('source_hash')}, now {source_hash(GRID_KERNEL_SOURCES)}): "
                      "rerun scripts/gpu_profile.sh")
    return (pmc["write_bytes_per_point"] * points_per_launch
            + pmc["fetch_bytes_per_segment_corrected"] * segments_per_launch), None


def usable_cores():
    """The CPU baseline's thread count: the CPUs this process may run on - the affinity mask, capped by the
    cgroup's CPU quota when there is one (the GPU boxes of this pool show 256 CPUs and allow 16 CPUs' worth of
    time per period: 256 threads then run in bursts between throttles, 0.3 % parallel efficiency, while 16 pinned
    threads keep 0.6-0.9). Returns (threads, how it was decided)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = f"{cores} CPUs in the affinity mask"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            allowed = max(1, int(quota) // int(period))
            if allowed < cores:
                cores, how = allowed, f"cgroup cpu.max {quota} {period} = {allowed} CPUs of the {cores} visible"
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and quota // period < cores:
                cores, how = max(1, quota // period), f"cgroup cfs quota {quota}/{period} of the {cores} visible"
        except (OSError, ValueError):
            pass
    return cores, how


def summary(seconds, units):
    """{median, min, max} rates of a repeated timing."""
    rates = sorted(units / s for s in seconds)
    return {"median": statistics.median(rates), "min": rates[0], "max": rates[-1], "repetitions": len(rates)}


def fit_kernel_name(kernel_ms):
    """The kernel that fitted the models of a call: the one of the k_fit_models family with the most time."""
    family = {name: ms for name, ms in kernel_ms.items() if name.startswith("k_fit_models")}
    return max(family, key=family.get) if family else "k_fit_models"


def spread(seconds):
    """{median, min, max, repetitions} of a repeated timing, in seconds."""
    return {"median": statistics.median(seconds), "min": min(seconds), "max": max(seconds), "repetitions": len(seconds)}


def timed_fit(context, call, repetitions):
    """`call()` (a fit that returns device segments) once untimed - the context's scratch grows -, then `repetitions`
    times under the wall clock with the kernels' HIP-event times; returns (segments of the last call, seconds[],
    {kernel: mean ms per call})."""
    call().free()
    context.sync()
    context.profile_enable(True)
    context.profile_reset()
    seconds, segments = [], None
    for _ in range(max(1, repetitions)):
        if segments is not None:
            segments.free()
        context.sync()
        started = time.perf_counter()
        segments = call()
        context.sync()
        seconds.append(time.perf_counter() - started)
    kernels = {name: ms / len(seconds) for name, (launches, ms) in context.profile().items() if name.startswith("k_fit")}
    context.profile_enable(False)
    return segments, seconds, kernels


def fit_on_gpu(context, mdb, np, args, rank, keep_series):
    """Generate this rank's series on the device and compress them with the HIP fitter, in groups of
    series so that raw values never need more than a few GB of HBM at once. Returns the device
    batches, the wall-clock seconds of every repetition (summed over the groups) with the mean HIP-event
    time of every fit kernel, and the raw values of the first `keep_series` series exactly as the fitter read them."""
    eb = mdb.error_bound("relative", args.error_bound)
    chunks_per_series = (args.points + CHUNK_POINTS - 1) // CHUNK_POINTS
    # All series of the rank in one launch when they fit (one lane per chunk: occupancy = chunks).
    group = max(1, min(args.series, args.fit_group_points // max(args.points, 1)))
    parts, fit_seconds, fit_points, kernel_ms, kept = [], None, 0, {}, None
    first_series_of_rank = rank * args.series
    for first in range(0, args.series, group):
        n_series = min(group, args.series - first)
        total = n_series * args.points
        values = context.dev_alloc(4 * total)
        context.synth_values_dev(values, first_series_of_rank + first, n_series, args.points, SEED)
        starts = np.arange(0, args.points, CHUNK_POINTS, dtype=np.uint64)
        offsets = (np.arange(n_series, dtype=np.uint64)[:, None] * np.uint64(args.points) + starts[None, :]).reshape(-1)
        offsets = np.concatenate([offsets, np.array([total], dtype=np.uint64)])
        first_index = np.tile(starts, n_series)
        k = n_series * chunks_per_series
        offsets_dev = context.upload_array(offsets)
        first_index_dev = context.upload_array(first_index)
        segments, seconds, kernels = timed_fit(
            context, lambda: context.compress_chunks_dev(0, values, offsets_dev, k, eb, 0, INTERVAL_US, first_index_dev),
            args.fit_repetitions)
        parts.append(segments)
        fit_seconds = seconds if fit_seconds is None else [a + b for a, b in zip(fit_seconds, seconds)]
        for name, ms in kernels.items():
            kernel_ms[name] = kernel_ms.get(name, 0.0) + ms
        fit_points += total
        if first == 0 and keep_series > 0:
            kept = context.download_array(values, min(keep_series, n_series) * args.points, np.float32)
        for pointer in (values, offsets_dev, first_index_dev):
            context.dev_free(pointer)
    return parts, fit_seconds, fit_points, kernel_ms, kept


def link_rates(local_rank=0, nbytes=1 << 30, repetitions=3):
    """What the box's PCIe link gives a plain copy between page-locked host memory and HBM, each way (GB/s, best of a few
    1-GiB copies on torch's stream): the yardstick for the host path's rates - the link's paper peak (PCIe 5.0 x16: 63
    GB/s) is not what a copy engine reaches."""
    import torch
    device = torch.device("cuda", local_rank)
    host_block = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    device_block = torch.empty(nbytes, dtype=torch.uint8, device=device)
    rates = {}
    for name, copy in (("h2d_GB_per_s", lambda: device_block.copy_(host_block, non_blocking=True)),
                       ("d2h_GB_per_s", lambda: host_block.copy_(device_block, non_blocking=True))):
        copy()
        torch.cuda.synchronize(device)
        best = float("inf")
        for _ in range(repetitions):
            started = time.perf_counter()
            copy()
            torch.cuda.synchronize(device)
            best = min(best, time.perf_counter() - started)
        rates[name] = nbytes / best / 1e9
    del host_block, device_block
    torch.cuda.empty_cache()
    return rates


def segment_bytes_up(batch):
    """What mdb_grid_submit sends to the device for these host segments: 57 B of fixed columns and three 16-byte views a
    row is not counted finer than that - 73 B a segment - plus the out-of-line payloads."""
    payload = sum(int(buffer.size) for column in (batch.timestamps, batch.values, batch.residuals) for buffer in column.buffers)
    return 73 * len(batch) + payload


def host_path(context, mdb, np, sample, args):
    """The drop-in path end to end: the C++ GridExec / GridStream of libmdb_host over HOST segment
    batches (what DataFusion would hand it), PCIe included: upload of the segments, kernels, copy of
    the reconstructed columns back into page-locked memory. Not the headline (`value` is
    device-resident); reported so the integrated number is on the record."""
    from modelardb_rs_amd import host
    chunks_per_series = (args.points + CHUNK_POINTS - 1) // CHUNK_POINTS
    sample_series = max(1, min(args.series, 1_000_000_000 // max(args.points, 1)))
    sample = sample.take(np.nonzero(sample.chunk_index < sample_series * chunks_per_series)[0])
    out = {"note": "the call sequence of the patched GridStream (rust/patches/0001-grid_exec.patch through "
                   "rust/modelardb_hip), issued by its C++ twin in libmdb_host: host segment batches of 8 192 rows "
                   "are gathered into mdb_grid_submit calls of about 16 M data points, one submit is kept ahead "
                   "(mdb_grid_wait of one while the next is on the GPU), tag views are repeated per row by the "
                   "library; polled to the end in slices of batch_size data points. Upload of the segments, kernels "
                   "and the copy of 12 B per data point into page-locked host memory included"}
    # The first pass over the sample is the cold one: the context's pool of page-locked blocks grows to the sizes
    # the batches need (a hipHostMalloc of 70 MB takes 13 ms). A server's pool is warm; both are reported.
    try:
        out["link"] = link_rates(context.device)
    except Exception as error:  # noqa: BLE001 - a yardstick, not a measurement of the path
        out["link"] = {"error": str(error)}
    d2h_rate = out["link"].get("d2h_GB_per_s")
    points, seconds, bytes_down = host.measure_grid_stream(context, sample, 8192)
    out["first_pass_cold_pool"] = {"values_per_s": points / seconds, "GB_per_s_pcie": bytes_down / seconds / 1e9}
    try:
        bytes_up = segment_bytes_up(sample)
    except Exception:  # noqa: BLE001
        bytes_up = None
    for batch_size in (8192, 65536):
        points, seconds, bytes_down = host.measure_grid_stream(context, sample, batch_size)
        out[f"batch_{batch_size}"] = {"values_per_s": points / seconds, "GB_per_s_pcie": bytes_down / seconds / 1e9,
                                      "points": points, "segments": len(sample), "seconds": seconds,
                                      "fraction_of_a_plain_d2h_copy": bytes_down / seconds / 1e9 / d2h_rate if d2h_rate else None,
                                      "GB_per_s_up": bytes_up / seconds / 1e9 if bytes_up else None}
    # With a tag column (grid_exec.rs:341-346): a 16-byte view per data point, written by host threads.
    host.measure_grid_stream(context, sample, 8192, tags={"tag": "wind-turbine-0042"})
    points, seconds, bytes_down = host.measure_grid_stream(context, sample, 8192, tags={"tag": "wind-turbine-0042"})
    out["batch_8192_one_tag_column"] = {"values_per_s": points / seconds, "GB_per_s_pcie": bytes_down / seconds / 1e9,
                                        "seconds": seconds}
    # What gathering buys: one input batch per submit (round 2's call shape) against the learned size.
    os.environ["MDB_HOST_GRID_COALESCE_SEGMENTS"] = "1"
    try:
        host.measure_grid_stream(context, sample, 8192)  # (the pool of page-locked blocks gets blocks of this size)
        points, seconds, bytes_down = host.measure_grid_stream(context, sample, 8192)
    finally:
        del os.environ["MDB_HOST_GRID_COALESCE_SEGMENTS"]
    out["batch_8192_one_input_batch_per_submit"] = {"values_per_s": points / seconds,
                                                    "GB_per_s_pcie": bytes_down / seconds / 1e9}
    # SUM through the patched accumulator (rust/patches/0002-model_simple_aggregates.patch): update_batch is handed
    # 8 192-row batches as by DataFusion and keeps them until 262 144 segments are pending or the state is read, then
    # ONE mdb_agg_batch_list.
    host.measure_accumulator(context, sample, host.ModelSumAccumulator)
    state, seconds = host.measure_accumulator(context, sample, host.ModelSumAccumulator)
    out["sum_accumulator_batch_8192"] = {"segments_per_s": len(sample) / seconds, "values_per_s": points / seconds,
                                         "seconds": seconds, "update_batch_calls": (len(sample) + 8191) // 8192,
                                         "library_calls": (len(sample) + 262143) // 262144, "sum": state[0]}
    # The middle half of the time axis (BASELINE configs 3 and 5) through the same operators.
    out["range_middle_half"] = host_range_rows(
        context, mdb, host, sample, (args.points // 4) * INTERVAL_US, (3 * args.points // 4) * INTERVAL_US,
        {"grid_seconds": out["batch_8192"]["seconds"], "grid_points": out["batch_8192"]["points"], "sum_seconds": seconds})
    return out


def host_range_rows(context, mdb, host, sample, t_lo, t_hi, whole):
    """BASELINE configs 3 and 5 through the operators: a query with `t_lo <= timestamp <= t_hi`. GridStream is handed
    the predicate TimeSeriesTable::scan makes of it, takes the range out (rust/patches/0001: time_range_of_predicate)
    and every mdb_grid_submit carries it; the SUM accumulator is the one the extended ModelSimpleAggregates rule
    creates for such a query (rust/patches/0002) and folds its batches with ONE mdb_agg_batch_range_list. `whole`:
    the same measurements without a range ({"grid_seconds", "grid_points", "sum_seconds"})."""
    import numpy as np
    predicate = f"(and (>= timestamp ts:{t_lo}) (<= timestamp ts:{t_hi}))"
    assert host.time_range_of_predicate(predicate) == (t_lo, t_hi, True)
    # What reaches the operators: TimeSeriesTable::scan also hands the range to the Parquet scan as a filter on the
    # segments (start_time <= t_hi AND end_time >= t_lo, query/time_series_table.rs:290-331, pushdown_filters = true), so
    # only segments with a point in the range are read at all. The filter is the scan's work, not the operators'.
    all_segments = len(sample)
    sample = sample.take(np.nonzero((sample.start_time <= t_hi) & (sample.end_time >= t_lo))[0])
    host.measure_grid_stream(context, sample, 8192, predicate=predicate)
    points, seconds, bytes_down = host.measure_grid_stream(context, sample, 8192, predicate=predicate)
    host.measure_accumulator(context, sample, host.ModelSumAccumulator, time_range=(t_lo, t_hi))
    state, sum_seconds = host.measure_accumulator(context, sample, host.ModelSumAccumulator, time_range=(t_lo, t_hi))
    count_state, _ = host.measure_accumulator(context, sample, host.ModelCountAccumulator, time_range=(t_lo, t_hi))
    if count_state[0] != points:
        raise SystemExit(f"VERIFICATION FAILED: host path under a time range: GridStream returned {points} data points, "
                         f"the COUNT accumulator counted {count_state[0]}")
    one_call = context.agg_batch_range(sample, t_lo, t_hi, mdb.MDB_AGG_SUM | mdb.MDB_AGG_COUNT)
    if one_call.count != points or abs(state[0] - one_call.sum) > 1e-9 * abs(one_call.sum):
        raise SystemExit(f"VERIFICATION FAILED: host path under a time range: SUM through the accumulator {state[0]!r}, "
                         f"through one call {one_call.sum!r}")
    return {"t_lo": t_lo, "t_hi": t_hi, "predicate": predicate, "segments_of_the_table": all_segments,
            "segments_the_parquet_filter_lets_through": len(sample),
            "grid": {"points": points, "fraction_of_points": points / max(whole["grid_points"], 1), "seconds": seconds,
                     "values_per_s": points / seconds, "GB_per_s_pcie": bytes_down / seconds / 1e9,
                     "seconds_over_unranged": seconds / whole["grid_seconds"]},
            "sum_accumulator": {"seconds": sum_seconds, "sum": state[0], "points": points,
                                "segments_per_s": len(sample) / sum_seconds,
                                "ms_over_unranged": (sum_seconds - whole["sum_seconds"]) * 1e3}}


def host_fit(context, mdb, np, ora, host_ts, values, offsets, eb, gpu_fitted):
    """The fit from host memory the way the patched call sites make it (rust/patches/0003, 0004): every series x
    field or finished buffer is one chunk where it lies, all of them in ONE mdb_compress_chunk_list - against the
    reference's call shape (one call per buffer) and, per launch size, against one CPU thread of the port."""
    n_chunks = len(offsets) - 1
    chunks = [(host_ts[int(a):int(b)], values[int(a):int(b)]) for a, b in zip(offsets[:-1], offsets[1:])]
    points = int(offsets[-1])
    context.compress_chunk_list(chunks, eb)
    from_host = context.compress_chunk_list(chunks, eb)
    seconds = context.last_call_seconds
    if from_host.rows() != gpu_fitted.rows():
        raise SystemExit("VERIFICATION FAILED: the fit through host pointers differs from the device-resident one")
    # Where such a call spends its time: a third call with the library's profile on (mdb_profile_get, "host:" names).
    context.profile_enable(True)
    context.profile_reset()
    context.compress_chunk_list(chunks, eb)
    phases = {name[len("host:chunk_list_"):] + "_ms": round(ms, 3) for name, (_, ms) in context.profile().items()
              if name.startswith("host:chunk_list_")}
    phases["kernels_ms"] = round(sum(ms for name, (_, ms) in context.profile().items() if not name.startswith("host:")), 3)
    phases["call_ms"] = round(1e3 * context.last_call_seconds, 3)
    context.profile_enable(False)
    context.profile_reset()
    out = {"points_per_s": points / seconds, "segments_per_s": len(from_host) / seconds, "seconds": seconds,
           "points": points, "chunks": n_chunks, "phases_of_a_profiled_call": phases,
           "note": "mdb_compress_chunk_list over the chunks where they lie in host memory (12 B per point handed "
                   "over; host threads gather the values into page-locked memory with streaming stores, slice by "
                   "slice while the previous slice crosses PCIe, and find every chunk's timestamps equally spaced, so "
                   "the timestamps never cross), segments downloaded; phases_of_a_profiled_call: gather = the threads' "
                   "copies (the slices' copies to the device run behind them), upload_tail = what of those is left "
                   "after the last slice, fit = kernels with their waits, download = the segments to the host; second of two calls; segments == the device-resident fit's"}
    # The reference's call shape: one call per finished buffer (uncompressed_data_manager.rs:505-596).
    per_buffer = chunks[:64]
    context.compress_chunk_list(per_buffer[:1], eb)
    started = time.perf_counter()
    for chunk in per_buffer:
        context.compress_chunk_list([chunk], eb)
    seconds = time.perf_counter() - started
    out["one_call_per_buffer"] = {"points_per_s": sum(len(v) for _, v in per_buffer) / seconds,
                                  "ms_per_buffer": 1e3 * seconds / len(per_buffer), "buffers": len(per_buffer)}
    # Latency by launch size against ONE CPU thread (the reference's single compression thread).
    cpu_chunks = min(n_chunks, 32)
    cpu_offsets = (offsets[:cpu_chunks + 1] - offsets[0]).astype(np.uint64)
    cpu_points = int(cpu_offsets[-1])
    _, cpu_seconds = ora.compress_chunks_timed(host_ts[:cpu_points], values[:cpu_points], cpu_offsets, eb, 1,
                                               repetitions=3)
    cpu_ms_per_chunk = 1e3 * statistics.median(cpu_seconds) / cpu_chunks
    table = []
    for n in (1, 4, 16, 64, 256, 1024, 4096):
        launch = [chunks[k % n_chunks] for k in range(n)]
        context.compress_chunk_list(launch, eb)
        timings = []
        for _ in range(3):
            context.compress_chunk_list(launch, eb)
            timings.append(context.last_call_seconds)
        gpu_ms = 1e3 * min(timings)
        table.append({"chunks": n, "points": sum(len(v) for _, v in launch), "gpu_ms": round(gpu_ms, 3),
                      "cpu_1_thread_ms": round(cpu_ms_per_chunk * n, 3),
                      "gpu_points_per_s": sum(len(v) for _, v in launch) / (gpu_ms * 1e-3)})
    out["fit_latency"] = {"rows": table, "cpu_ms_per_chunk": cpu_ms_per_chunk,
                          "note": "chunks of 65 536 points (the server's ingest buffer) from host memory through "
                                  "mdb_compress_chunk_list to segments in host memory, best of 3; the CPU column is "
                                  "the port on one thread (measured on 32 chunks, scaled)"}
    return out


def irregular_timestamps(context, mdb, np, args):
    """Series whose timestamps are materialised and NOT equally spaced - the delta-of-delta streams of
    timestamps.rs:228-292: fit, grid(), COUNT/MIN/MAX/SUM on the segments and the same under WHERE timestamp
    BETWEEN, of 100 of the benchmark's series (10^9 points at the default size), for timestamps spaced at random and
    for a fixed rate with one sample in a hundred missing, next to the same series with regular timestamps. Not
    the headline; reported so that the cost of irregular timestamps is on the record of every run."""
    series = max(1, min(args.series, 100, 1_000_000_000 // max(args.points, 1)))
    points, total = args.points, series * args.points
    eb = mdb.error_bound("relative", args.error_bound)
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    values = context.dev_alloc(4 * total)
    context.synth_values_dev(values, 0, series, points)
    starts = np.arange(0, points, CHUNK_POINTS, dtype=np.uint64)
    offsets = np.concatenate([(s * points + starts) for s in range(series)] + [np.array([total], dtype=np.uint64)]).astype(np.uint64)
    offsets_dev = context.upload_array(offsets)
    rng = np.random.default_rng(5)
    shapes = (("regular", np.arange(points, dtype=np.int64) * 1000),
              ("random_intervals", np.cumsum(rng.integers(900, 1100, points).astype(np.int64))),
              ("one_percent_gaps", np.cumsum(np.where(rng.random(points) < 0.01, 2000, 1000).astype(np.int64))))
    out = {"points": total, "series": series,
           "note": "timestamps materialised on the device; randomly spaced (every delta different) and a fixed "
                   "rate with 1 % of the samples missing, next to the same series equally spaced; grid and "
                   "aggregates: mean of 3 calls on resident segments after a first one, which is timed by itself (a "
                   "resident batch keeps what the walk of its timestamp streams found, and for the aggregates under "
                   "a time range what a pass over the whole time axis made of every segment: later calls walk only "
                   "the segments their range cuts); the first million reconstructed timestamps "
                   "and the counts are checked"}
    for label, one in shapes:
        timestamps = np.tile(one, series)
        ts_dev = context.upload_array(timestamps)
        segments, fit_seconds, fit_kernels = timed_fit(
            context, lambda: context.compress_chunks_dev(ts_dev, values, offsets_dev, len(offsets) - 1, eb, 0, 0, 0),
            args.fit_repetitions)
        n = context.grid_count_dev(segments)
        if n != total:
            raise SystemExit(f"VERIFICATION FAILED: {label}: {n} points in the segments, {total} fitted")
        out_ts, out_val = context.dev_alloc(8 * n), context.dev_alloc(4 * n)
        t_lo, t_hi = int(one[points // 4]), int(one[3 * points // 4])
        calls = (("grid", lambda: context.grid_batch_dev(segments, out_ts, out_val, n)),
                 ("aggregates", lambda: context.agg_batch_dev(segments, mask)),
                 ("aggregates_between_quartiles", lambda: context.agg_batch_range_dev(segments, t_lo, t_hi, mask)))
        shape = {"fit_ms": 1e3 * statistics.median(fit_seconds), "fit_ms_min": 1e3 * min(fit_seconds),
                 "fit_ms_max": 1e3 * max(fit_seconds), "fit_repetitions": len(fit_seconds),
                 "fit_kernels_ms": {k: round(v, 3) for k, v in fit_kernels.items() if v > 0.05}, "segments": len(segments)}
        for name, call in calls:
            context.sync(); started = time.perf_counter()
            call()
            context.sync()
            # (the first call leaves what its walk of the timestamp streams found with the resident batch)
            shape[name + "_first_call_ms"] = 1e3 * (time.perf_counter() - started)
            context.profile_enable(True); context.profile_reset(); context.sync(); started = time.perf_counter()
            for _ in range(3):
                result = call()
            context.sync(); seconds = (time.perf_counter() - started) / 3
            shape[name + "_ms"] = 1e3 * seconds
            shape[name + "_kernels_ms"] = {k: round(v[1] / v[0], 3) for k, v in context.profile().items() if v[1] / v[0] > 0.05}
            context.profile_enable(False)
            if name == "aggregates" and result.count != total:
                raise SystemExit(f"VERIFICATION FAILED: {label}: COUNT {result.count} of {total} points")
            if name == "aggregates_between_quartiles":
                expected = series * int(np.count_nonzero((one >= t_lo) & (one <= t_hi)))
                if result.count != expected:
                    raise SystemExit(f"VERIFICATION FAILED: {label}: COUNT {result.count} BETWEEN, expected {expected}")
        if not np.array_equal(context.download_array(out_ts, min(n, 1_000_000), np.int64), timestamps[:min(n, 1_000_000)]):
            raise SystemExit(f"VERIFICATION FAILED: {label}: reconstructed timestamps differ from the ones fitted")
        out[label] = shape
        for pointer in (ts_dev, out_ts, out_val):
            context.dev_free(pointer)
        segments.free()
        del timestamps
    context.dev_free(values)
    context.dev_free(offsets_dev)
    return out


def verified_aggregates(context, mdb, np, ora, sample, t_lo, t_hi):
    """COUNT / MIN / MAX / SUM of the sample segments, plain and under WHERE timestamp BETWEEN, from the HIP library
    against the oracle: COUNT, MIN and MAX exact (bit patterns), SUM within the reference's own 0.001 %
    (crates/modelardb_server/tests/integration_test.rs:1128-1171)."""
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    bits = lambda x: int(np.float32(x).view(np.uint32))
    checked = {}
    for label, got, expected in (
            ("plain", context.agg_batch(sample, mask), ora.agg_batch(sample, mask)),
            ("range", context.agg_batch_range(sample, t_lo, t_hi, mask), ora.agg_batch_range(sample, t_lo, t_hi, mask))):
        if (got.count, bits(got.min), bits(got.max)) != (expected.count, bits(expected.min), bits(expected.max)):
            raise SystemExit(f"VERIFICATION FAILED: {label} aggregates: COUNT/MIN/MAX {got.count, got.min, got.max} "
                             f"but the oracle has {expected.count, expected.min, expected.max}")
        if abs(got.sum - expected.sum) > 1e-5 * abs(expected.sum):
            raise SystemExit(f"VERIFICATION FAILED: {label} aggregates: SUM {got.sum} but the oracle has {expected.sum}")
        checked[label] = {"count": got.count, "sum_relative_difference": abs(got.sum - expected.sum) / max(abs(expected.sum), 1e-300)}
    return checked


def mixed_models(context, mdb, np, ora, args):
    """The other model types on the record of every run: the reference's own acceptance recipe
    (crates/modelardb_compression/src/compression.rs:733-863 over crates/modelardb_test/src/data_generation.rs:
    108-284: runs of 50..500 points that are Constant, Linear or Random in 100..200, half of the series with noise
    from 1.0..1.05 added, regular timestamps 100 us apart) at 10^9 points, under a lossless and a relative 1 % bound:
    fit, grid and segment aggregates with their rates against the HBM peak, the segment mix, and the first series
    fitted and reconstructed by the oracle."""
    import datagen
    points = 1_000_000
    distinct = max(2, min(64, args.mixed_points // points // 2 * 2))
    copies = max(1, args.mixed_points // (distinct * points))
    series, total = distinct * copies, distinct * copies * points
    host_values = np.concatenate([datagen.mixed_series(points, 1000 + s, (1.0, 1.05) if s % 2 else None)[1]
                                  for s in range(distinct)])
    values = context.dev_alloc(4 * total)
    for copy in range(copies):
        context.lib.mdb_dev_upload(context.handle, values + 4 * copy * distinct * points,
                                   host_values.ctypes.data, host_values.nbytes)
    starts = np.arange(0, points, CHUNK_POINTS, dtype=np.uint64)
    offsets = np.concatenate([(s * points + starts) for s in range(series)] + [np.array([total], dtype=np.uint64)]).astype(np.uint64)
    first_index = np.tile(starts, series)
    offsets_dev, first_index_dev = context.upload_array(offsets), context.upload_array(first_index)
    n_chunks = len(offsets) - 1
    chunks_per_series = len(starts)
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    sample_ts = np.arange(points, dtype=np.int64) * 100
    out = {"points": total, "series": series, "distinct_series": distinct,
           "note": "the reference's acceptance recipe (compression.rs:733-863): runs of 50..500 points, Constant / Linear "
                   "/ Random(100..200), every second series with noise 1.0..1.05 added, regular timestamps; chunks of "
                   "65 536 points; fit: median of 5 calls after a first one; grid and aggregates (all points; the middle half of the "
                   "time axis): median of 5 calls on the resident segments; bytes: 4 B/point read by the fit, 73 B/segment + out-of-line payloads + 12 B/point for "
                   "grid, 73 B/segment + out-of-line payloads for the aggregates; checked: the first two series "
                   "(one without, one with noise) fitted by the oracle == the GPU's segments, their grid == the "
                   "oracle's, COUNT == points; COUNT / MIN / MAX of the two series' segments, plain and BETWEEN the quartiles, == "
                   "the oracle's (bit patterns), SUM within 0.001 %"}
    for label, eb in (("lossless", mdb.error_bound("lossless")), ("relative_1_percent", mdb.error_bound("relative", 1.0))):
        segments, fit_timings, fit_kernels = timed_fit(
            context, lambda: context.compress_chunks_dev(0, values, offsets_dev, n_chunks, eb, 0, 100, first_index_dev),
            args.fit_repetitions)
        fit_seconds = statistics.median(fit_timings)
        fit_kernels = {k: round(v, 3) for k, v in fit_kernels.items() if v > 0.05}
        n = context.grid_count_dev(segments)
        if n != total:
            raise SystemExit(f"VERIFICATION FAILED: mixed models, {label}: {n} points in the segments, {total} fitted")
        seg = segments.seg
        payload_bytes = sum(int(col.buffer_sizes[b]) for col in (seg.timestamps, seg.values, seg.residuals)
                            for b in range(col.n_buffers))
        out_ts, out_val = context.dev_alloc(8 * n), context.dev_alloc(4 * n)
        shape = {"segments": len(segments), "out_of_line_payload_bytes": payload_bytes,
                 "fit": {"ms": 1e3 * fit_seconds, "ms_min": 1e3 * min(fit_timings), "ms_max": 1e3 * max(fit_timings),
                         "repetitions": len(fit_timings), "points_per_s": total / fit_seconds, "kernels_ms": fit_kernels,
                         "GB_per_s": 4.0 * total / fit_seconds / 1e9, "frac_of_hbm_peak": 4.0 * total / fit_seconds / 1e9 / HBM_PEAK_GBPS}}
        # (WHERE timestamp BETWEEN the quartiles of the series' common time axis, N1: half of every series' points)
        t_lo, t_hi = int(sample_ts[points // 4]), int(sample_ts[3 * points // 4])
        between = series * int(np.count_nonzero((sample_ts >= t_lo) & (sample_ts <= t_hi)))
        for name, call, algorithmic in (
                ("grid", lambda: context.grid_batch_dev(segments, out_ts, out_val, n), 73.0 * len(segments) + payload_bytes + 12.0 * n),
                ("aggregates", lambda: context.agg_batch_dev(segments, mask), 73.0 * len(segments) + payload_bytes),
                ("aggregates_between_quartiles", lambda: context.agg_batch_range_dev(segments, t_lo, t_hi, mask),
                 73.0 * len(segments) + payload_bytes / 2.0)):
            result = call()
            # The call as a caller gets it, then - its kernels bracketed by events one by one - where its time goes.
            timings = []
            for _ in range(5):
                context.sync(); started = time.perf_counter()
                result = call()
                context.sync(); timings.append(time.perf_counter() - started)
            seconds = statistics.median(timings)
            context.profile_enable(True); context.profile_reset()
            profiled = []
            for _ in range(3):
                context.sync(); started = time.perf_counter()
                call()
                context.sync(); profiled.append(time.perf_counter() - started)
            shape[name] = {"ms": 1e3 * seconds, "values_per_s": n / seconds, "GB_per_s": algorithmic / seconds / 1e9,
                           "frac_of_hbm_peak": algorithmic / seconds / 1e9 / HBM_PEAK_GBPS,
                           "ms_kernels_timed_one_by_one": 1e3 * statistics.median(profiled),
                           "kernels_ms": {k: round(v[1] / v[0], 3) for k, v in context.profile().items() if v[1] / v[0] > 0.05}}
            context.profile_enable(False)
            if name == "grid":
                shape["segment_mix"] = result[1]
            elif name == "aggregates_between_quartiles":
                if result.count != between:
                    raise SystemExit(f"VERIFICATION FAILED: mixed models, {label}: COUNT {result.count} BETWEEN, expected {between}")
            elif result.count != total:
                raise SystemExit(f"VERIFICATION FAILED: mixed models, {label}: COUNT {result.count} of {total} points")
            else:
                resident_sum = result.sum
        # the first two series (without and with noise) against the oracle: segments byte for byte, points bit for bit
        downloaded = segments.download()
        for s in range(2):
            gpu_segments = downloaded.take(np.nonzero((downloaded.chunk_index >= s * chunks_per_series) &
                                                      (downloaded.chunk_index < (s + 1) * chunks_per_series))[0])
            series_values = host_values[s * points:(s + 1) * points]
            series_offsets = np.concatenate([starts, [points]]).astype(np.uint64)
            expected = ora.compress_chunks(sample_ts, series_values, series_offsets, eb)
            if not gpu_segments.identical(expected):
                raise SystemExit(f"VERIFICATION FAILED: mixed models, {label}: the segments of series {s} differ from the oracle's")
            expected_ts, expected_values = ora.grid_batch(expected)[:2]
            got_ts = context.download_array(out_ts, points, np.int64, offset_elements=s * points)
            got_values = context.download_array(out_val, points, np.float32, offset_elements=s * points)
            if not (np.array_equal(got_ts, expected_ts) and np.array_equal(got_values.view(np.uint32), expected_values.view(np.uint32))):
                raise SystemExit(f"VERIFICATION FAILED: mixed models, {label}: the grid of series {s} differs from the oracle's")
        two_series = downloaded.take(np.nonzero(downloaded.chunk_index < 2 * chunks_per_series)[0])
        shape["aggregates_verified"] = verified_aggregates(context, mdb, np, ora, two_series, t_lo, t_hi)
        shape["aggregates_verified"]["sample"] = f"the first two series ({len(two_series)} segments)"
        if not args.no_host_path:
            # The same segments as host batches through the pipelined grid of the boundary (the patched GridStream's
            # call sequence, PCIe included): all three model types, MacaqueV streams without the resident batch's
            # cursors.
            from modelardb_rs_amd import host
            host.measure_grid_stream(context, downloaded, 8192)
            host_points, host_seconds, host_bytes = host.measure_grid_stream(context, downloaded, 8192)
            shape["host_path"] = {"values_per_s": host_points / host_seconds, "GB_per_s_pcie": host_bytes / host_seconds / 1e9,
                                  "seconds": host_seconds, "segments": len(downloaded)}
            try:  # (the link carries the segments one way while the points come back the other)
                shape["host_path"]["GB_per_s_up"] = segment_bytes_up(downloaded) / host_seconds / 1e9
            except Exception:  # noqa: BLE001
                pass
            # SUM through the patched accumulator (rust/patches/0002): 8 192-row batches gathered into mdb_agg_batch_list.
            host.measure_accumulator(context, downloaded, host.ModelSumAccumulator)
            state, sum_seconds = host.measure_accumulator(context, downloaded, host.ModelSumAccumulator)
            if abs(state[0] - resident_sum) > 1e-9 * abs(resident_sum):
                raise SystemExit(f"VERIFICATION FAILED: mixed models, {label}: SUM through the accumulator {state[0]!r}, "
                                 f"of the resident segments {resident_sum!r}")
            shape["host_path"]["sum_accumulator"] = {"values_per_s": total / sum_seconds, "segments_per_s": len(downloaded) / sum_seconds,
                                                     "seconds": sum_seconds, "sum": state[0]}
            shape["host_path"]["range_middle_half"] = host_range_rows(
                context, mdb, host, downloaded, t_lo, t_hi,
                {"grid_seconds": host_seconds, "grid_points": host_points, "sum_seconds": sum_seconds})
        del downloaded
        out[label] = shape
        for pointer in (out_ts, out_val):
            context.dev_free(pointer)
        segments.free()
    for pointer in (values, offsets_dev, first_index_dev):
        context.dev_free(pointer)
    return out


def fit_counters():
    """Vector / scalar instructions per data point and wave of the fit's model kernel from the committed SQ counter
    passes (bench.py cannot collect counters itself), valid while the fit kernel's source is the one measured."""
    path = os.path.join(REPO_ROOT, "profiles", "pmc_fit_models.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_fit_models.json is missing"
    with open(path) as f:
        pmc = json.load(f)
    if pmc.get("source_hash") != source_hash(FIT_KERNEL_SOURCES):
        return None, (f"the fit kernel sources changed since the SQ counter passes (measured {pmc.get('source_hash')}, "
                      f"now {source_hash(FIT_KERNEL_SOURCES)}): rerun scripts/r05/pmc_fit.sh")
    return pmc, None


def fit_roofline(kernel_ms, points, info):
    """Both bounds of the model kernel of a fit: the HBM figure the contract names (4 B per point read: regular
    timestamps are synthesised) and the one that binds - vector instruction issue: a wave of 64 chunks issues
    `valu_per_point` vector instructions per step of 64 points, 4 cycles each on its SIMD, so the floor of a launch is
    points / 64 x valu_per_point x 4 cycles / (SIMDs x clock)."""
    name = fit_kernel_name(kernel_ms)
    model_ms = sum(ms for kernel, ms in kernel_ms.items() if kernel.startswith("k_fit_models") or kernel == "k_fit_walk")
    gbps = 4.0 * points / (model_ms * 1e-3) / 1e9 if model_ms > 0 else 0.0
    out = {"bound": "hbm", "kernel": name, "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": gbps / HBM_PEAK_GBPS, "kernel_ms": model_ms, "algorithmic_bytes_per_launch": 4.0 * points,
           "note": "4 B/point read (regular timestamps are synthesised, not loaded); the greedy fit is a sequential "
                   "dependency per chunk - HBM is the contract's bound, not the one that binds: see valu_issue"}
    pmc, why_not = fit_counters()
    if pmc is None or pmc.get("kernel") != name:
        out["valu_issue"] = None
        out["valu_issue_note"] = why_not or f"the SQ counter passes measured {pmc.get('kernel')}, this launch ran {name}"
        return out
    simds = 4 * int(info.get("compute_units", 256))
    clock_hz = SHADER_CLOCK_MHZ * 1e6
    floor_ms = 1e3 * (points / 64.0) * pmc["valu_per_point"] * 4.0 / (simds * clock_hz)
    out["valu_issue"] = {"valu_per_point": pmc["valu_per_point"], "salu_per_point": pmc.get("salu_per_point"),
                         "simds": simds, "clock_mhz": SHADER_CLOCK_MHZ, "floor_ms": floor_ms,
                         "frac": floor_ms / model_ms if model_ms > 0 else 0.0,
                         "note": "vector instructions per step of one wave (64 chunks, one point each) from "
                                 "profiles/pmc_fit_models.json (SQ_INSTS_VALU / steps), 4 cycles per instruction and "
                                 "wave on a SIMD; frac = floor / measured kernel time"}
    return out


def fit_check_and_cpu(args, np, mdb, ora, rank, cores, fit_sample_values, n_fit, gpu_fitted):
    """The oracle's greedy compression of the very bytes the GPU fitted (the first `n_fit` series of the rank): the
    segments must be identical in every column; the same calls, timed, are the CPU baseline of the fit (the port on
    `cores` pinned threads, chunks sharded, and on one thread). Also checks the device generator against its host
    definition. Returns (cpu baseline, oracle segments, host timestamps, chunk offsets, generator points checked)."""
    import datagen
    eb = mdb.error_bound("relative", args.error_bound)
    host_ts = np.tile(np.arange(args.points, dtype=np.int64) * INTERVAL_US, n_fit)
    offsets = np.array([s * args.points + c for s in range(n_fit)
                        for c in range(0, args.points, CHUNK_POINTS)] + [n_fit * args.points], dtype=np.uint64)
    host_defined = np.concatenate([datagen.bench_series(rank * args.series + s, min(args.points, 1 << 20), SEED)
                                   for s in range(n_fit)])
    device_made = np.concatenate([fit_sample_values[s * args.points: s * args.points + min(args.points, 1 << 20)]
                                  for s in range(n_fit)])
    if not np.array_equal(host_defined.view(np.uint32), device_made.view(np.uint32)):
        raise SystemExit("VERIFICATION FAILED: the device generator differs from tests/datagen.bench_series")
    fitted, fit_cpu_seconds = ora.compress_chunks_timed(host_ts, fit_sample_values, offsets, eb, cores, repetitions=3)
    single_chunks = min(len(offsets) - 1, 64)
    single_offsets = offsets[:single_chunks + 1]
    _, fit_single_seconds = ora.compress_chunks_timed(host_ts[:int(single_offsets[-1])], fit_sample_values[:int(single_offsets[-1])],
                                                      single_offsets, eb, 1, repetitions=3)
    if not fitted.identical(gpu_fitted):
        raise SystemExit("VERIFICATION FAILED: GPU segments differ from the oracle's")
    fit_rates = summary(fit_cpu_seconds, n_fit * args.points)
    fit_single_rate = summary(fit_single_seconds, int(single_offsets[-1]))["median"]
    fit_cpu = {"value": fit_rates["median"], "unit": "points/s",
               "points_per_s": fit_rates["median"], "min": fit_rates["min"], "max": fit_rates["max"],
               "repetitions": fit_rates["repetitions"],
               "segments_per_s": len(fitted) / statistics.median(fit_cpu_seconds), "cores": cores,
               "kind": "port", "single_thread_points_per_s": fit_single_rate,
               "parallel_efficiency": fit_rates["median"] / cores / fit_single_rate if fit_single_rate > 0 else None,
               "sample": f"the fit of {n_fit} series x {args.points} points of the same workload (the oracle's "
                         f"try_compress_univariate_time_series per 65 536-point chunk), chunks sharded over a standing "
                         f"pool of {cores} pinned threads; median of 3 timed passes"}
    return fit_cpu, fitted, host_ts, offsets, len(host_defined)


PHASES = {}


class phase:
    """Wall time of one part of the run, into the line's `phases_s` and onto stderr as it ends."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.started = time.perf_counter()

    def __exit__(self, *exc):
        PHASES[self.name] = PHASES.get(self.name, 0.0) + time.perf_counter() - self.started
        print(f"[bench] {self.name}: {PHASES[self.name]:.2f} s", file=sys.stderr, flush=True)


def claim_stdout():
    """The contract is ONE JSON line on stdout. RCCL prints a version banner to the C-level stdout
    when a communicator is created (flushed at exit, i.e. after the line), so file descriptor 1 is
    pointed at stderr for the whole run and the line is written to the original descriptor."""
    sys.stdout.flush()
    original = os.dup(1)
    os.dup2(2, 1)
    return original


class GpuWorkload:
    """BASELINE configs[1] on one rank's GPU: build = fit on the GPU (setup), step = grid() of all resident
    segments into resident columns, report = roofline of the dominant kernel from HIP events, the segment
    aggregates merged over RCCL (a collective: every rank), and - on rank 0 only - the CPU legs, the
    verification against the oracle and the secondary blocks."""

    def __init__(self, args, rank, local_rank, world, dist):
        import numpy as np
        import modelardb_rs_amd as mdb
        from modelardb_rs_amd import sharding
        self.args, self.rank, self.local_rank, self.world, self.dist = args, rank, local_rank, world, dist
        self.np, self.mdb = np, mdb
        self.context = mdb.Context(local_rank)
        self.info = self.context.device_info()
        with phase("comm_init"):
            sharding.init_comm(self.context, dist)  # the C ABI's own RCCL communicator (mdb_comm_init)
        self.verify = rank == 0 and not args.no_cpu_baseline
        self.parts, self.out_ts, self.out_val = [], None, None

    def sync(self):
        import torch
        self.context.sync()
        torch.cuda.synchronize()

    def build(self):
        args, context, np = self.args, self.context, self.np
        self.n_fit_sample = min(args.fit_sample_series, args.series) if self.verify else 0
        with phase("generate_and_fit"):
            (self.parts, self.fit_seconds, self.fit_points, self.fit_kernel_ms,
             self.fit_sample_values) = fit_on_gpu(context, self.mdb, np, args, self.rank, self.n_fit_sample)
        self.n_segments = sum(len(p) for p in self.parts)
        self.total_points = sum(context.grid_count_dev(part) for part in self.parts)
        assert self.total_points == args.series * args.points, (self.total_points, args.series * args.points)
        self.out_ts = context.dev_alloc(8 * self.total_points)
        self.out_val = context.dev_alloc(4 * self.total_points)
        self.ranged = 0.0 < args.range_middle < 1.0
        self.step_lo = int(args.points * (0.5 - args.range_middle / 2)) * INTERVAL_US
        self.step_hi = int(args.points * (0.5 + args.range_middle / 2)) * INTERVAL_US
        settle_until = time.perf_counter() + args.settle_seconds
        while time.perf_counter() < settle_until:
            self.step()
            context.sync()

    def step(self):
        context, at, metrics_total = self.context, 0, None
        for part in self.parts:
            if self.ranged:
                n, metrics = context.grid_batch_range_dev(part, self.step_lo, self.step_hi, self.out_ts + 8 * at,
                                                          self.out_val + 4 * at, self.total_points - at)
            else:
                n, metrics = context.grid_batch_dev(part, self.out_ts + 8 * at, self.out_val + 4 * at,
                                                    self.total_points - at)
            at += n
            if metrics_total is None:
                metrics_total = dict(metrics)
            else:
                for key, value in metrics.items():
                    metrics_total[key] += value
        self.produced, self.metrics = at, metrics_total
        return at

    def report(self, elapsed, per_rank_seconds):
        args, context, np, mdb, dist = self.args, self.context, self.np, self.mdb, self.dist
        rank, world, parts = self.rank, self.world, self.parts
        ranged, step_lo, step_hi = self.ranged, self.step_lo, self.step_hi
        total_points, n_segments, out_ts, out_val = self.total_points, self.n_segments, self.out_ts, self.out_val
        assert ranged or self.produced == total_points
        points_per_step = self.produced
        metrics = self.metrics

        # ---- roofline of the dominant kernel: HIP events on the launch stream ------------------------
        context.profile_enable(True)
        context.profile_reset()
        for _ in range(args.steps):
            self.step()
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
        traffic, traffic_note = pmc_traffic(args, points_per_launch, segments_per_launch)

        # ---- the segment aggregates (BASELINE config 3) on the same resident segments: median of 5 -----
        mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
        t_lo, t_hi = (args.points // 4) * INTERVAL_US, (3 * args.points // 4) * INTERVAL_US
        payload_bytes = sum(int(col.buffer_sizes[b]) for part in parts
                            for col in (part.seg.timestamps, part.seg.values, part.seg.residuals)
                            for b in range(col.n_buffers))
        agg_bytes = 73.0 * n_segments + payload_bytes

        def timed_aggregates(call):
            for part in parts:  # warm
                call(part, None)
            context.profile_enable(True)
            context.profile_reset()
            seconds, state = [], None
            for _ in range(5):
                context.sync()
                started = time.perf_counter()
                state = None
                for part in parts:
                    state = call(part, state)
                context.sync()
                seconds.append(time.perf_counter() - started)
            kernels = {name: ms / max(n, 1) for name, (n, ms) in context.profile().items()}
            context.profile_enable(False)
            return state, seconds, kernels

        state, agg_seconds, agg_kernels = timed_aggregates(lambda part, st: context.agg_batch_dev(part, mask, st))
        range_state, range_seconds, range_kernels = timed_aggregates(
            lambda part, st: context.agg_batch_range_dev(part, t_lo, t_hi, mask, st))
        # The one exchange step of the path: merge the per-GPU aggregate partials with the C ABI's
        # mdb_agg_all_reduce (one 32-byte ncclAllGather over RCCL / xGMI + a rank-ordered fold).
        local_state = mdb._abi.AggStateC(state.sum, state.count, state.min, state.max)
        context.agg_all_reduce(state)  # first collective on the communicator: connection set-up
        t0 = time.perf_counter()
        state, ranks_seen = context.agg_all_reduce(local_state)
        range_state, _ = context.agg_all_reduce(range_state)
        reduce_seconds = (time.perf_counter() - t0) / 2
        assert ranks_seen == world, (ranks_seen, world)
        assert state.count == world * total_points, (state.count, world * total_points)
        self.ranks_seen = ranks_seen

        # Under a time range a resident batch keeps what a pass over the whole time axis made of every segment
        # (DESIGN.md section 3c): the timed calls read start and end time of every segment (16 B), the kept 24 B of the
        # ones the range contains (here: half of them) and the whole row only of the ones it cuts (two per series).
        range_bytes = 16.0 * n_segments + 24.0 * n_segments * ((t_hi - t_lo) / (args.points * INTERVAL_US))

        def aggregate_roofline(seconds, kernels, name, bytes_read=None, note=None):
            median = statistics.median(seconds)
            kernel = kernels.get(name, 0.0)
            bytes_read = agg_bytes if bytes_read is None else bytes_read
            return {"bound": "hbm", "kernel": name, "achieved": bytes_read / (kernel * 1e-3) / 1e9 if kernel > 0 else 0.0,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": bytes_read / (kernel * 1e-3) / 1e9 / HBM_PEAK_GBPS if kernel > 0 else 0.0,
                    "kernel_ms": kernel, "call_ms": {"median": 1e3 * median, "min": 1e3 * min(seconds), "max": 1e3 * max(seconds)},
                    "algorithmic_bytes_per_launch": bytes_read,
                    "note": note or ("73 B per segment (the columns a SUM reads) + out-of-line payloads, output O(1); "
                                     "kernel time from HIP events over 5 calls, call time = wall clock of the calls")}

        aggregates = {
            "segments_per_s": n_segments / statistics.median(agg_seconds),
            "seconds": statistics.median(agg_seconds),
            "kernel_ms": agg_kernels.get("k_agg_segments", 0.0),
            "roofline": aggregate_roofline(agg_seconds, agg_kernels, "k_agg_segments"),
            "result": {"count": state.count, "min": state.min, "max": state.max, "sum": state.sum,
                       "avg": state.sum / max(state.count, 1)},
            "range": {"t_lo": t_lo, "t_hi": t_hi, "seconds": statistics.median(range_seconds),
                      "segments_per_s": n_segments / statistics.median(range_seconds),
                      "kernel_ms": range_kernels.get("k_agg_range", 0.0),
                      "roofline": aggregate_roofline(
                          range_seconds, range_kernels, "k_agg_range", range_bytes,
                          "16 B per segment (start and end time) + 24 B per segment the range contains (what the batch "
                          "keeps of it: sum, count, min, max; built by the first call, which is not among the timed ones) "
                          "- not the 73 B row, which only the segments the range cuts are read for; kernel time from "
                          "HIP events over 5 calls, call time = wall clock of the calls"),
                      "count": range_state.count, "min": range_state.min, "max": range_state.max, "sum": range_state.sum},
            "final_reduce_seconds": reduce_seconds,
            "note": "COUNT/MIN/MAX/SUM on the resident segments (BASELINE config 3: no grid); the range "
                    "variant clips to the middle half of the time axis; the partials of all ranks are merged "
                    "by mdb_agg_all_reduce (C ABI): one 32-byte all-gather over RCCL per merge",
        }
        if rank != 0:
            return None

        # ---- rank 0's tail: in-run verification + CPU baseline (the oracle on a bounded sample) ---------
        cpu_baseline = fit_cpu = verified = host_path_result = irregular_result = mixed_result = None
        secondary = world == 1 or args.full_tail
        if self.verify:
            import oracle_lib as ora
            cores, cores_how = usable_cores()
            eb = mdb.error_bound("relative", args.error_bound)
            chunks_per_series = (args.points + CHUNK_POINTS - 1) // CHUNK_POINTS
            with phase("download_segments"):
                downloaded = parts[0].download()
            n_sample = min(args.cpu_sample_series, args.series)
            # Whole leading series: segments are ordered by chunk, chunks by series.
            sample = downloaded.take(np.nonzero(downloaded.chunk_index < n_sample * chunks_per_series)[0])
            with phase("cpu_baseline_grid"):
                ts_cpu, val_cpu, cpu_seconds = ora.grid_batch_timed(sample, cores, repetitions=3)
                single = downloaded.take(np.nonzero(downloaded.chunk_index < chunks_per_series)[0])
                ts_single, _, single_seconds = ora.grid_batch_timed(single, 1, repetitions=3)
            rates = summary(cpu_seconds, len(ts_cpu))
            single_rate = summary(single_seconds, len(ts_single))["median"]
            cpu_baseline = {
                "value": rates["median"], "min": rates["min"], "max": rates["max"],
                "repetitions": rates["repetitions"],
                "unit": "values/s",
                "cores": cores,
                "kind": "port",
                "sample": f"grid() of the first {n_sample} series ({len(ts_cpu)} points, {len(sample)} "
                          f"segments) of the same workload, segment ranges sharded over {cores} host "
                          f"threads; median of 3 timed passes after one untimed pass",
                "threads": f"a standing pool of {cores} worker threads ({cores_how}) made by the untimed pass, worker w "
                           "pinned to the w-th allowed CPU; NUMA: Linux first-touch, every worker first-touches "
                           "(untimed pass) the output pages it writes",
                "single_thread_value": single_rate,
                "parallel_efficiency": rates["median"] / cores / single_rate if single_rate > 0 else None,
            }
            # grid verification: the device columns of the timed step against the oracle's, bit for bit.
            if ranged:
                keep = (ts_cpu >= step_lo) & (ts_cpu <= step_hi)
                ts_cpu, val_cpu = ts_cpu[keep], val_cpu[keep]
            grid_points_verified = 0
            piece = 1 << 26
            verify_started = time.perf_counter()
            for at in range(0, len(ts_cpu), piece):
                n_piece = min(piece, len(ts_cpu) - at)
                got_ts = context.download_array(out_ts, n_piece, np.int64, offset_elements=at)
                got_val = context.download_array(out_val, n_piece, np.float32, offset_elements=at)
                if not np.array_equal(got_ts, ts_cpu[at:at + n_piece]):
                    raise SystemExit("VERIFICATION FAILED: grid timestamps differ from the oracle")
                if not np.array_equal(got_val.view(np.uint32), val_cpu[at:at + n_piece].view(np.uint32)):
                    raise SystemExit("VERIFICATION FAILED: grid values differ from the oracle")
                grid_points_verified += n_piece
            del ts_cpu, val_cpu
            # ... and at full size, a property that does not need the oracle: every one of the rank's series occupies
            # exactly its block of the output, from its first to its last (visible) timestamp.
            first_us = -(-step_lo // INTERVAL_US) * INTERVAL_US if ranged else 0
            last_us = min(step_hi // INTERVAL_US, args.points - 1) * INTERVAL_US if ranged else (args.points - 1) * INTERVAL_US
            per_series = (last_us - first_us) // INTERVAL_US + 1
            if points_per_step != args.series * per_series:
                raise SystemExit(f"VERIFICATION FAILED: {points_per_step} points per step, expected {args.series * per_series}")
            for series_index in range(args.series):
                block = series_index * per_series
                edge = (int(context.download_array(out_ts, 1, np.int64, offset_elements=block)[0]),
                        int(context.download_array(out_ts, 1, np.int64, offset_elements=block + per_series - 1)[0]))
                if edge != (first_us, last_us):
                    raise SystemExit(f"VERIFICATION FAILED: series {series_index} starts / ends at {edge}")
            PHASES["verify_grid"] = time.perf_counter() - verify_started
            print(f"[bench] verify_grid: {PHASES['verify_grid']:.2f} s", file=sys.stderr, flush=True)
            with phase("verify_aggregates"):
                # (a sample of the segments through the same kernels, against the oracle's per-row loops)
                n_agg = min(8, n_sample)
                agg_sample = downloaded.take(np.nonzero(downloaded.chunk_index < n_agg * chunks_per_series)[0])
                aggregates["verified"] = verified_aggregates(context, mdb, np, ora, agg_sample, t_lo, t_hi)
                aggregates["verified"]["sample"] = f"the first {n_agg} series ({len(agg_sample)} segments)"
            verify_started = time.perf_counter()
            # fit verification: the oracle's greedy compression of the very bytes the GPU fitted.
            n_fit = self.n_fit_sample
            fit_sample_values = self.fit_sample_values
            gpu_fitted = downloaded.take(np.nonzero(downloaded.chunk_index < n_fit * chunks_per_series)[0])
            fit_cpu, fitted, host_ts, offsets, host_defined = fit_check_and_cpu(
                args, np, mdb, ora, rank, cores, fit_sample_values, n_fit, gpu_fitted)
            verified = {"fit_segments": len(fitted), "fit_points": n_fit * args.points,
                        "grid_points": grid_points_verified, "generator_points": host_defined,
                        "series_blocks": args.series, "aggregates": aggregates["verified"],
                        "how": "after the timed region: oracle fit of the sample series' exact bytes == GPU "
                               "segments (all columns, bit patterns); oracle grid of the sample == the device "
                               "columns the timed step wrote (bit for bit); COUNT/MIN/MAX of a sample of the segments, "
                               "plain and over the time range, == the oracle's, SUM within 0.001 %; device generator == "
                               "host definition; every series of the rank starts and ends its block of the output "
                               "columns where it must"}
            PHASES["verify_fit_and_cpu_baseline_fit"] = time.perf_counter() - verify_started
            print(f"[bench] verify_fit_and_cpu_baseline_fit: {PHASES['verify_fit_and_cpu_baseline_fit']:.2f} s", file=sys.stderr, flush=True)
            if secondary and not args.no_host_path:
                with phase("host_path"):
                    host_path_result = host_path(context, mdb, np, downloaded, args)
                with phase("host_path_fit"):
                    host_path_result["fit"] = host_fit(context, mdb, np, ora, host_ts, fit_sample_values, offsets, eb,
                                                       gpu_fitted)
            del downloaded
            if secondary and not args.no_irregular:
                with phase("irregular_timestamps"):
                    irregular_result = irregular_timestamps(context, mdb, np, args)
            if secondary and not args.no_mixed_models:
                # (the headline's columns and segments are no longer needed: room for 10^9 points of other data)
                with phase("mixed_models"):
                    mixed_result = mixed_models(context, mdb, np, ora, args)

        value = world * points_per_step * args.steps / elapsed
        fit_kernel_ms, fit_points, fit_timings = self.fit_kernel_ms, self.fit_points, self.fit_seconds
        fit_seconds = statistics.median(fit_timings)
        return {
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
            "rccl_ranks_seen": ranks_seen,
            "ms_per_step_per_rank": {"min": 1e3 * min(per_rank_seconds) / args.steps,
                                     "max": 1e3 * max(per_rank_seconds) / args.steps},
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
                "arithmetic": "Swing values as (f64 slope * f64 t + f64 intercept) -> f32, timestamps i64; "
                              "output columns i64 + f32 (12 B/point)",
                "device": self.info["name"],
                "libraries": loaded_runtime_libraries(),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_grid_tiles",
                "achieved": achieved_gbps,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved_gbps / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": algorithmic_bytes,
                "other_kernels_ms": {name: ms / max(n, 1) for name, (n, ms) in profile.items()
                                     if name != "k_grid_tiles"},
            },
            "cpu_baseline": cpu_baseline,
            "verified": verified,
            "phases_s": {name: round(seconds, 2) for name, seconds in PHASES.items()},
            "aggregates": aggregates,
            "host_path": host_path_result,
            "irregular_timestamps": irregular_result,
            "mixed_models": mixed_result,
            "fit": {
                "cpu_baseline": fit_cpu,
                "points_per_s": fit_points / fit_seconds if fit_seconds > 0 else None,
                "segments_per_s": n_segments / fit_seconds if fit_seconds > 0 else None,
                "seconds": fit_seconds,
                "seconds_spread": spread(fit_timings),
                "roofline": fit_roofline(fit_kernel_ms, fit_points, self.info),
                "kernels_ms": fit_kernel_ms,
                "note": "PMC-Mean/Swing/MacaqueV fit of this rank's series on the GPU (setup, not "
                        "in the timed region; `--timed fit` makes it the timed step), regular timestamps synthesised "
                        "on the fly; median of the repetitions after one untimed call",
            },
        }

    def free_headline_buffers(self):
        for part in self.parts:
            part.free()
        self.parts = []
        for pointer in (self.out_ts, self.out_val):
            if pointer:
                self.context.dev_free(pointer)
        self.out_ts = self.out_val = None

    def close(self):
        self.free_headline_buffers()
        self.context.close()  # (mdb_comm_close first, then the stream and the scratch)


class FitWorkload:
    """BASELINE configs[3] (the compression path) on one rank's GPU, `--timed fit`: build = this rank's series
    generated into HBM, step = ONE mdb_compress_chunks_dev over all of them (PMC-Mean / Swing / MacaqueV fit of every
    65 536-point chunk, segments written as Arrow columns in HBM; the previous step's segments are freed first),
    report = both rooflines of the model kernel from HIP events, and - on rank 0 - the CPU baseline (the fit port)
    and the verification of a sample of the series against the oracle. Series shard over the ranks (weak scaling),
    no data-path collective; the aggregates of the fitted segments are merged over RCCL once, as in the grid run."""

    def __init__(self, args, rank, local_rank, world, dist):
        import numpy as np
        import modelardb_rs_amd as mdb
        from modelardb_rs_amd import sharding
        self.args, self.rank, self.local_rank, self.world, self.dist = args, rank, local_rank, world, dist
        self.np, self.mdb = np, mdb
        self.context = mdb.Context(local_rank)
        self.info = self.context.device_info()
        with phase("comm_init"):
            sharding.init_comm(self.context, dist)
        self.verify = rank == 0 and not args.no_cpu_baseline
        self.values = self.offsets_dev = self.first_index_dev = self.segments = None

    def sync(self):
        import torch
        self.context.sync()
        torch.cuda.synchronize()

    def build(self):
        args, context, np = self.args, self.context, self.np
        if args.series * args.points > args.fit_group_points:
            raise SystemExit(f"--timed fit keeps the rank's {args.series * args.points} points resident in one launch; "
                             f"raise --fit-group-points (now {args.fit_group_points}) or lower --series / --points")
        self.eb = self.mdb.error_bound("relative", args.error_bound)
        self.total = args.series * args.points
        with phase("generate"):
            self.values = context.dev_alloc(4 * self.total)
            context.synth_values_dev(self.values, self.rank * args.series, args.series, args.points, SEED)
            starts = np.arange(0, args.points, CHUNK_POINTS, dtype=np.uint64)
            offsets = (np.arange(args.series, dtype=np.uint64)[:, None] * np.uint64(args.points) + starts[None, :]).reshape(-1)
            offsets = np.concatenate([offsets, np.array([self.total], dtype=np.uint64)])
            self.n_chunks = len(offsets) - 1
            self.chunks_per_series = len(starts)
            self.offsets_dev = context.upload_array(offsets)
            self.first_index_dev = context.upload_array(np.tile(starts, args.series))
            context.sync()
        with phase("first_fit"):  # (the context's scratch grows: tens of GB of hipMalloc)
            self.step()
            context.sync()

    def step(self):
        if self.segments is not None:
            self.segments.free()
        self.segments = self.context.compress_chunks_dev(0, self.values, self.offsets_dev, self.n_chunks, self.eb, 0,
                                                         INTERVAL_US, self.first_index_dev)
        return self.total

    def report(self, elapsed, per_rank_seconds):
        args, context, np, mdb = self.args, self.context, self.np, self.mdb
        rank, world = self.rank, self.world
        n_segments = len(self.segments)
        # ---- the kernels of a step: HIP events on the launch stream ------------------------------------
        context.profile_enable(True)
        context.profile_reset()
        call_seconds = []
        for _ in range(args.steps):
            context.sync(); started = time.perf_counter()
            self.step()
            context.sync(); call_seconds.append(time.perf_counter() - started)
        kernel_ms = {name: ms / args.steps for name, (launches, ms) in context.profile().items() if name.startswith("k_fit")}
        context.profile_enable(False)
        if context.grid_count_dev(self.segments) != self.total:
            raise SystemExit("VERIFICATION FAILED: the fitted segments do not hold the points that were fitted")
        # ---- the one exchange step of the path, on the fitted segments ----------------------------------
        mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
        state = context.agg_batch_dev(self.segments, mask)
        context.agg_all_reduce(state)
        state, ranks_seen = context.agg_all_reduce(mdb._abi.AggStateC(state.sum, state.count, state.min, state.max))
        assert ranks_seen == world and state.count == world * self.total, (ranks_seen, state.count)
        if rank != 0:
            return None
        cpu_baseline = verified = None
        if self.verify:
            import oracle_lib as ora
            cores, cores_how = usable_cores()
            n_fit = min(args.fit_sample_series, args.series)
            with phase("download_sample"):
                sample_values = context.download_array(self.values, n_fit * args.points, np.float32)
                downloaded = self.segments.download()
                gpu_fitted = downloaded.take(np.nonzero(downloaded.chunk_index < n_fit * self.chunks_per_series)[0])
                del downloaded
            with phase("verify_fit_and_cpu_baseline_fit"):
                cpu_baseline, fitted, _, _, generator_points = fit_check_and_cpu(
                    args, np, mdb, ora, rank, cores, sample_values, n_fit, gpu_fitted)
                cpu_baseline["threads"] = f"a standing pool of {cores} worker threads ({cores_how}), worker w pinned to the w-th allowed CPU"
            verified = {"fit_segments": len(fitted), "fit_points": n_fit * args.points, "generator_points": generator_points,
                        "grid_count": self.total,
                        "how": "after the timed region: the oracle's fit of the sample series' exact bytes == the segments "
                               "of the last timed step (all columns, bit patterns); the segments of the rank hold exactly "
                               "the points that were fitted (mdb_grid_count); device generator == host definition"}
        value = world * self.total * args.steps / elapsed
        mix = context.grid_count_dev(self.segments)
        return {
            "metric": "fit points/sec",
            "value": value,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "rccl_ranks_seen": ranks_seen,
            "segments_per_s": world * n_segments * args.steps / elapsed,
            "ms_per_step_per_rank": {"min": 1e3 * min(per_rank_seconds) / args.steps,
                                     "max": 1e3 * max(per_rank_seconds) / args.steps},
            "config": {
                "workload": f"{args.series} series x {args.points} points sine+noise per GPU, relative error bound "
                            f"{args.error_bound} %, PMC-Mean / Swing / MacaqueV fit of every {CHUNK_POINTS}-point chunk "
                            f"(values resident in HBM, regular timestamps, segments written as Arrow columns in HBM)",
                "series_per_gpu": args.series,
                "points_per_series": args.points,
                "chunks_per_gpu": self.n_chunks,
                "segments_per_gpu": n_segments,
                "points_in_segments": mix,
                "parallelism": f"series-sharded x{world}, no data-path collective",
                "arithmetic": "Swing in f64 (no contraction), PMC-Mean f64 sum / f32 tests, MacaqueV u32 bit streams",
                "device": self.info["name"],
                "libraries": loaded_runtime_libraries(),
            },
            "roofline": fit_roofline(kernel_ms, self.total, self.info),
            "kernels_ms": {name: round(ms, 3) for name, ms in kernel_ms.items()},
            "call_ms": {"median": 1e3 * statistics.median(call_seconds), "min": 1e3 * min(call_seconds),
                        "max": 1e3 * max(call_seconds)},
            "cpu_baseline": cpu_baseline,
            "verified": verified,
            "aggregates_of_fitted_segments": {"count": state.count, "min": state.min, "max": state.max, "sum": state.sum},
            "phases_s": {name: round(seconds, 2) for name, seconds in PHASES.items()},
        }

    def close(self):
        if self.segments is not None:
            self.segments.free()
            self.segments = None
        for pointer in (self.values, self.offsets_dev, self.first_index_dev):
            if pointer:
                self.context.dev_free(pointer)
        self.values = self.offsets_dev = self.first_index_dev = None
        self.context.close()


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "rccl_ranks_seen")
COMPACT_LIMIT_BYTES = 4096
DETAIL_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_detail.json")


def _dig(tree, *path):
    """tree[path[0]][path[1]]... or None wherever a step is missing."""
    for key in path:
        if not isinstance(tree, dict) or key not in tree:
            return None
        tree = tree[key]
    return tree


def _short(value, digits=5):
    """Floats to `digits` significant figures (the detail file keeps them whole), strings cut at 300 characters."""
    if isinstance(value, bool) or value is None:
        return value
    if isinstance(value, float):
        if value != value or value in (float("inf"), float("-inf")):
            return None  # (strict JSON has no NaN / Infinity)
        return float(f"{value:.{digits}g}")
    if isinstance(value, str):
        return value if len(value) <= 300 else value[:297] + "..."
    if isinstance(value, dict):
        return {key: _short(item, digits) for key, item in value.items()}
    if isinstance(value, (list, tuple)):
        return [_short(item, digits) for item in value]
    return value


def loaded_runtime_libraries():
    """Which HIP / RCCL files this process has mapped (/proc/self/maps): the first thing to look at when an N > 1 run
    misbehaves (torch bundles its own copies next to /opt/rocm's)."""
    seen = {}
    try:
        with open("/proc/self/maps") as maps:
            for row in maps:
                path = row.rstrip("\n").rpartition(" ")[2]
                base = os.path.basename(path)
                for stem in ("libamdhip64", "librccl", "libhsa-runtime64", "libmdb_hip", "libmdb_host"):
                    if base.startswith(stem):
                        seen.setdefault(stem, set()).add(path)
    except OSError:
        return None
    return {stem: sorted(paths) for stem, paths in sorted(seen.items())}


def compact_line(line, detail_name="bench_detail.json"):
    """The ONE line the driver parses: the contract's keys, `config` (what the workload was), `roofline` and
    `cpu_baseline` of the timed kernel, and a dozen scalars of the secondary blocks under `also` - everything else is
    in the detail file. Kept under COMPACT_LIMIT_BYTES (asserted here and in tests/test_bench_cpu.py): the 21.9 KB
    line of round 5 was more than the driver reads back."""
    out = {key: line[key] for key in CONTRACT_KEYS if key in line}
    config = line.get("config") or {}
    out["config"] = {key: config[key] for key in ("workload", "series_per_gpu", "points_per_series", "chunks_per_gpu",
                                                  "segments_per_gpu", "segment_mix", "parallelism", "device",
                                                  "libraries") if key in config}
    roofline = line.get("roofline")
    if isinstance(roofline, dict):
        out["roofline"] = {key: roofline[key] for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                          "kernel_ms", "algorithmic_bytes_per_launch") if key in roofline}
        if isinstance(roofline.get("valu_issue"), dict):  # (the fit's own bound: vector issue, not HBM)
            out["roofline"]["valu_issue_frac"] = roofline["valu_issue"].get("frac")
    else:
        out["roofline"] = roofline
    cpu = line.get("cpu_baseline")
    if isinstance(cpu, dict):
        out["cpu_baseline"] = {key: cpu[key] for key in ("value", "unit", "cores", "kind", "single_thread_value", "sample")
                               if key in cpu}
    else:
        out["cpu_baseline"] = cpu
    also = {
        # the other rows of the path, each next to the roofline that bounds it (fractions of the 8 TB/s HBM peak)
        "fit_points_per_s": _dig(line, "fit", "points_per_s"),
        "fit_kernel": _dig(line, "fit", "roofline", "kernel"),
        "fit_kernel_ms": _dig(line, "fit", "roofline", "kernel_ms"),
        "fit_frac_of_hbm": _dig(line, "fit", "roofline", "frac"),
        "fit_frac_of_valu_issue": _dig(line, "fit", "roofline", "valu_issue", "frac"),
        "fit_cpu_points_per_s": _dig(line, "fit", "cpu_baseline", "value"),
        "aggregates_frac_of_hbm": _dig(line, "aggregates", "roofline", "frac"),
        "range_aggregates_frac_of_hbm": _dig(line, "aggregates", "range", "roofline", "frac"),
        # the series with all three model types (MacaqueV decode runs HERE, not in the headline's segments)
        "mixed_grid_frac_lossless": _dig(line, "mixed_models", "lossless", "grid", "frac_of_hbm_peak"),
        "mixed_grid_frac_1pct": _dig(line, "mixed_models", "relative_1_percent", "grid", "frac_of_hbm_peak"),
        "mixed_grid_ms_lossless": _dig(line, "mixed_models", "lossless", "grid", "ms"),
        "mixed_grid_ms_1pct": _dig(line, "mixed_models", "relative_1_percent", "grid", "ms"),
        "mixed_aggregates_frac_lossless": _dig(line, "mixed_models", "lossless", "aggregates", "frac_of_hbm_peak"),
        "mixed_aggregates_frac_1pct": _dig(line, "mixed_models", "relative_1_percent", "aggregates", "frac_of_hbm_peak"),
        "mixed_aggregates_ms_lossless": _dig(line, "mixed_models", "lossless", "aggregates", "ms"),
        "mixed_aggregates_ms_1pct": _dig(line, "mixed_models", "relative_1_percent", "aggregates", "ms"),
        "mixed_fit_ms_lossless": _dig(line, "mixed_models", "lossless", "fit", "ms"),
        "mixed_fit_ms_1pct": _dig(line, "mixed_models", "relative_1_percent", "fit", "ms"),
        "host_path_values_per_s": _dig(line, "host_path", "batch_8192", "values_per_s"),
        "host_path_GB_per_s_pcie": _dig(line, "host_path", "batch_8192", "GB_per_s_pcie"),
        "host_path_fraction_of_plain_d2h": _dig(line, "host_path", "batch_8192", "fraction_of_a_plain_d2h_copy"),
        "segments_per_s": line.get("segments_per_s"),
        "aggregates_count": _dig(line, "aggregates", "result", "count"),
    }
    out["also"] = {key: value for key, value in also.items() if value is not None}
    verified = line.get("verified")
    if isinstance(verified, dict):
        out["verified"] = {key: value for key, value in verified.items() if isinstance(value, (int, float))}
    out["ms_per_step_per_rank"] = line.get("ms_per_step_per_rank")
    out["detail"] = detail_name
    out = _short(out, 6)
    for key in ("value", "ms_per_step"):  # (the contract's two numbers stay whole)
        if key in line:
            out[key] = line[key]
    text = json.dumps(out, allow_nan=False)
    if len(text) > COMPACT_LIMIT_BYTES:
        raise SystemExit(f"bench.py's line is {len(text)} bytes, over the {COMPACT_LIMIT_BYTES} it is held to")
    return text


def control_group(dist, backend):
    """The group the ranks' checkpoints run on: gloo over loopback next to an RCCL job (a rank that waits for rank 0's
    tail sleeps in a socket instead of spinning on a stream, and a checkpoint can never be mismatched with a
    collective of the data path), the job's own group when that is gloo already."""
    if backend != "nccl":
        return None
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    return dist.new_group(backend="gloo")


def orchestrate(args, make_workload, backend="nccl", result_fd=1):
    """The rank protocol of the contract, for any workload object with build / step / sync / report / close:
    W untimed steps, a barrier + device sync, K timed steps, a barrier + device sync, the MAX over ranks; then the
    report (its collectives on every rank, rank 0's tail alone) and ONE meeting point: all ranks close their
    communicators and the process group in the same order, rank 0 prints the line only if every rank was fine, and
    every rank exits non-zero otherwise.
    Every rank issues the SAME sequence of collectives whatever happens to it: each phase ends in a checkpoint - an
    all-reduce (MIN) of "I am fine" on the control group, which is also the barrier of the contract - and a rank
    that has failed keeps taking part in the checkpoints with 0, so that all ranks skip the rest together instead
    of meeting in different collectives. A rank that dies is noticed through the process group's timeout."""
    import datetime
    import traceback
    with phase("init_distributed"):
        rank, local_rank, world, dist = init_distributed(args, backend, datetime.timedelta(seconds=args.collective_timeout))
    import torch
    state = {"failure": None}
    try:
        control = control_group(dist, backend)
    except BaseException as error:  # noqa: BLE001
        traceback.print_exc(file=sys.stderr)
        control, state["failure"] = None, error

    def attempt(what):
        """Run `what()` unless this rank has failed already; a failure is reported and remembered."""
        if state["failure"] is not None:
            return None
        try:
            return what()
        except BaseException as error:  # noqa: BLE001 - reported, then the job ends non-zero on every rank
            traceback.print_exc(file=sys.stderr)
            state["failure"] = error
            return None

    def checkpoint():
        """True if every rank is fine. (Also a barrier.)"""
        fine = torch.tensor([0 if state["failure"] is not None else 1], dtype=torch.int32)
        if backend == "nccl" and control is None:
            fine = fine.to(torch.device("cuda", local_rank))
        try:
            dist.all_reduce(fine, op=dist.ReduceOp.MIN, group=control)
            return bool(fine.item())
        except BaseException as error:  # noqa: BLE001 - a rank is gone: leave with an error, do not hang
            traceback.print_exc(file=sys.stderr)
            state["failure"] = state["failure"] or error
            return False

    workload_box, line = [None], None

    def build_and_warm_up():
        workload_box[0] = make_workload(args, rank, local_rank, world, dist)
        workload_box[0].build()
        for _ in range(args.warmup):
            workload_box[0].step()
        workload_box[0].sync()

    def timed_steps():
        for _ in range(args.steps):
            workload_box[0].step()
        workload_box[0].sync()

    attempt(build_and_warm_up)
    everyone_fine = checkpoint()                  # barrier + (above) device sync in front of the timed region
    if everyone_fine:
        attempt(lambda: workload_box[0].sync())
        t0 = time.perf_counter()
        attempt(timed_steps)                      # K steps, then the device sync
        everyone_fine = checkpoint()              # the barrier behind it
        attempt(lambda: workload_box[0].sync())
        own_elapsed = time.perf_counter() - t0
    if everyone_fine:
        # MAX over ranks (and every rank's own time, for the min/max on the line).
        t = torch.tensor([own_elapsed], dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        attempt(lambda: dist.all_gather(gathered, t, group=control))
        everyone_fine = checkpoint()
    if everyone_fine:
        per_rank_seconds = [float(x.item()) for x in gathered]
        line = attempt(lambda: workload_box[0].report(max(per_rank_seconds), per_rank_seconds))
    # The meeting point (the ranks other than 0 have been here since their report() returned).
    everyone_fine = checkpoint() and everyone_fine
    try:
        if workload_box[0] is not None:
            workload_box[0].close()
        dist.destroy_process_group()
    except BaseException:  # noqa: BLE001
        traceback.print_exc(file=sys.stderr)
        everyone_fine = False
    if rank == 0 and everyone_fine and line is not None:
        # Everything that was measured: into the detail file (and onto stderr); the compact line ALONE on stdout.
        detail_path = getattr(args, "detail_file", None) or DETAIL_FILE
        detail = json.dumps(line)
        try:
            with open(detail_path, "w") as sink:
                sink.write(detail + "\n")
        except OSError as error:
            print(f"[bench] could not write {detail_path}: {error}", file=sys.stderr, flush=True)
        print(f"[bench] detail: {detail}", file=sys.stderr, flush=True)
        os.write(result_fd, (compact_line(line, os.path.basename(detail_path)) + "\n").encode())
    return 0 if everyone_fine else 1


def main():
    args = parse_args()
    launch_ranks_if_needed(args)
    result_fd = claim_stdout()
    sys.exit(orchestrate(args, FitWorkload if args.timed == "fit" else GpuWorkload, "nccl", result_fd))


if __name__ == "__main__":
    main()
