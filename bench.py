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
    parser.add_argument("--range-middle", type=float, default=0.0,
                        help="BASELINE config 5: the timed step is a point-range GridExec query over this "
                             "fraction of the time axis (centred), e.g. 0.5; 0 = the whole series (config 2)")
    parser.add_argument("--settle-seconds", type=float, default=0.0,
                        help="run the step untimed for this long before the warmup (lets clocks settle)")
    parser.add_argument("--fit-group-points", type=int, default=13_000_000_000,
                        help="at most this many raw points (4 B each) are resident per fit launch")
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
    child = subprocess.run(command, env=environment, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(child.stdout)
    sys.stdout.flush()
    sys.exit(child.returncode)


def init_distributed(args):
    """Every run is a torch.distributed job over RCCL, the single-GPU one included (a process group
    of one rank), so the collective path is exercised wherever the bench runs."""
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
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
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


def barrier_and_sync(context, dist):
    """Both sides of the timed region: this rank's launch stream drained, every rank arrived, and
    (the barrier is a collective on torch's stream) torch's streams drained too."""
    import torch
    context.sync()
    dist.barrier()
    torch.cuda.synchronize()
    context.sync()


def summary(seconds, units):
    """{median, min, max} rates of a repeated timing."""
    rates = sorted(units / s for s in seconds)
    return {"median": statistics.median(rates), "min": rates[0], "max": rates[-1], "repetitions": len(rates)}


def fit_on_gpu(context, mdb, np, args, rank, keep_series):
    """Generate this rank's series on the device and compress them with the HIP fitter, in groups of
    series so that raw values never need more than a few GB of HBM at once. Returns the device
    batches, the timing of the (second, warm) fit call of every group with its k_fit_models kernel
    time, and the raw values of the first `keep_series` series exactly as the fitter read them."""
    eb = mdb.error_bound("relative", args.error_bound)
    chunks_per_series = (args.points + CHUNK_POINTS - 1) // CHUNK_POINTS
    # All series of the rank in one launch when they fit (one lane per chunk: occupancy = chunks).
    group = max(1, min(args.series, args.fit_group_points // max(args.points, 1)))
    parts, fit_seconds, fit_points, kernel_ms, kept = [], 0.0, 0, {}, None
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
        # The first call grows the context's scratch (tens of GB of hipMalloc); time the second.
        context.compress_chunks_dev(0, values, offsets_dev, k, eb, 0, INTERVAL_US, first_index_dev).free()
        context.sync()
        context.profile_enable(True)
        context.profile_reset()
        t0 = time.perf_counter()
        parts.append(context.compress_chunks_dev(0, values, offsets_dev, k, eb, 0, INTERVAL_US,
                                                 first_index_dev))
        context.sync()
        fit_seconds += time.perf_counter() - t0
        for name, (launches, ms) in context.profile().items():
            if name.startswith("k_fit"):
                kernel_ms[name] = kernel_ms.get(name, 0.0) + ms
        context.profile_enable(False)
        fit_points += total
        if first == 0 and keep_series > 0:
            kept = context.download_array(values, min(keep_series, n_series) * args.points, np.float32)
        for pointer in (values, offsets_dev, first_index_dev):
            context.dev_free(pointer)
    return parts, fit_seconds, fit_points, kernel_ms, kept


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
    points, seconds, bytes_down = host.measure_grid_stream(context, sample, 8192)
    out["first_pass_cold_pool"] = {"values_per_s": points / seconds, "GB_per_s_pcie": bytes_down / seconds / 1e9}
    for batch_size in (8192, 65536):
        points, seconds, bytes_down = host.measure_grid_stream(context, sample, batch_size)
        out[f"batch_{batch_size}"] = {"values_per_s": points / seconds, "GB_per_s_pcie": bytes_down / seconds / 1e9,
                                      "points": points, "segments": len(sample), "seconds": seconds}
    # With a tag column (grid_exec.rs:341-346): a 16-byte view per data point, written by host threads.
    host.measure_grid_stream(context, sample, 8192, tags={"tag": "wind-turbine-0042"})
    points, seconds, bytes_down = host.measure_grid_stream(context, sample, 8192, tags={"tag": "wind-turbine-0042"})
    out["batch_8192_one_tag_column"] = {"values_per_s": points / seconds, "GB_per_s_pcie": bytes_down / seconds / 1e9,
                                        "seconds": seconds}
    # What gathering buys: one input batch per submit (round 2's call shape) against the learned size.
    os.environ["MDB_HOST_GRID_COALESCE_SEGMENTS"] = "1"
    try:
        points, seconds, bytes_down = host.measure_grid_stream(context, sample, 8192)
    finally:
        del os.environ["MDB_HOST_GRID_COALESCE_SEGMENTS"]
    out["batch_8192_one_input_batch_per_submit"] = {"values_per_s": points / seconds,
                                                    "GB_per_s_pcie": bytes_down / seconds / 1e9}
    return out


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
    out = {"points_per_s": points / seconds, "segments_per_s": len(from_host) / seconds, "seconds": seconds,
           "points": points, "chunks": n_chunks,
           "note": "mdb_compress_chunk_list over the chunks where they lie in host memory (12 B per point handed "
                   "over; host threads gather the values into page-locked memory slice by slice while the previous "
                   "slice crosses PCIe and find every chunk's timestamps equally spaced, so the timestamps never "
                   "cross), segments downloaded; second of two calls; segments == the device-resident fit's"}
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
                   "aggregates: mean of 3 calls on resident segments; the first million reconstructed timestamps "
                   "and the counts are checked"}
    for label, one in shapes:
        timestamps = np.tile(one, series)
        ts_dev = context.upload_array(timestamps)
        context.compress_chunks_dev(ts_dev, values, offsets_dev, len(offsets) - 1, eb, 0, 0, 0).free()
        context.sync(); started = time.perf_counter()
        segments = context.compress_chunks_dev(ts_dev, values, offsets_dev, len(offsets) - 1, eb, 0, 0, 0)
        context.sync(); fit_seconds = time.perf_counter() - started
        n = context.grid_count_dev(segments)
        if n != total:
            raise SystemExit(f"VERIFICATION FAILED: {label}: {n} points in the segments, {total} fitted")
        out_ts, out_val = context.dev_alloc(8 * n), context.dev_alloc(4 * n)
        t_lo, t_hi = int(one[points // 4]), int(one[3 * points // 4])
        calls = (("grid", lambda: context.grid_batch_dev(segments, out_ts, out_val, n)),
                 ("aggregates", lambda: context.agg_batch_dev(segments, mask)),
                 ("aggregates_between_quartiles", lambda: context.agg_batch_range_dev(segments, t_lo, t_hi, mask)))
        shape = {"fit_ms": 1e3 * fit_seconds, "segments": len(segments)}
        for name, call in calls:
            call()
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


def main():
    args = parse_args()
    launch_ranks_if_needed(args)
    result_fd = claim_stdout()
    with phase("init_distributed"):
        rank, local_rank, world, dist = init_distributed(args)
    import numpy as np
    import torch

    import modelardb_rs_amd as mdb
    from modelardb_rs_amd import sharding

    context = mdb.Context(local_rank)
    info = context.device_info()
    with phase("comm_init"):
        sharding.init_comm(context, dist)  # the C ABI's own RCCL communicator (mdb_comm_init)
    verify = rank == 0 and not args.no_cpu_baseline

    # ---- build the workload: fit on the GPU, keep the segments in HBM ---------------------------
    n_fit_sample = min(args.fit_sample_series, args.series) if verify else 0
    with phase("generate_and_fit"):
        parts, fit_seconds, fit_points, fit_kernel_ms, fit_sample_values = fit_on_gpu(
            context, mdb, np, args, rank, n_fit_sample)
    n_segments = sum(len(p) for p in parts)

    total_points = 0
    for part in parts:
        total_points += context.grid_count_dev(part)
    assert total_points == args.series * args.points, (total_points, args.series * args.points)
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
    own_elapsed = time.perf_counter() - t0
    assert ranged or produced == total_points
    points_per_step = produced

    # MAX over ranks (and every rank's own time, for the min/max on the line).
    t = torch.tensor([own_elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    per_rank_seconds = [float(x.item()) for x in gathered]
    elapsed = max(per_rank_seconds)

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
    traffic, traffic_note = pmc_traffic(args, points_per_launch, segments_per_launch)

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
                "variant clips to the middle half of the time axis; the partials of all ranks are merged "
                "by mdb_agg_all_reduce (C ABI): one 32-byte all-gather over RCCL per merge",
    }

    # ---- in-run verification + CPU baseline: the oracle on a bounded sample -----------------------
    cpu_baseline = None
    fit_cpu = None
    verified = None
    host_path_result = None
    irregular_result = None
    if verify:
        import oracle_lib as ora
        cores = os.cpu_count() or 1
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
        cpu_baseline = {
            "value": rates["median"], "min": rates["min"], "max": rates["max"],
            "repetitions": rates["repetitions"],
            "unit": "values/s",
            "cores": cores,
            "kind": "port",
            "sample": f"grid() of the first {n_sample} series ({len(ts_cpu)} points, {len(sample)} "
                      f"segments) of the same workload, segment ranges sharded over {cores} host "
                      f"threads; median of 3 timed passes after one untimed pass",
            "threads": f"{cores} worker threads, worker w pinned to the w-th allowed CPU; NUMA: Linux "
                       "first-touch, every worker first-touches (untimed pass) the output pages it writes",
            "single_thread_value": summary(single_seconds, len(ts_single))["median"],
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
        verify_started = time.perf_counter()
        # fit verification: the oracle's greedy compression of the very bytes the GPU fitted.
        n_fit = n_fit_sample
        host_ts = np.tile(np.arange(args.points, dtype=np.int64) * INTERVAL_US, n_fit)
        offsets = np.array([s * args.points + c for s in range(n_fit)
                            for c in range(0, args.points, CHUNK_POINTS)] + [n_fit * args.points],
                           dtype=np.uint64)
        import datagen
        host_defined = np.concatenate([datagen.bench_series(rank * args.series + s, min(args.points, 1 << 20), SEED)
                                       for s in range(n_fit)])
        device_made = np.concatenate([fit_sample_values[s * args.points: s * args.points + min(args.points, 1 << 20)]
                                      for s in range(n_fit)])
        if not np.array_equal(host_defined.view(np.uint32), device_made.view(np.uint32)):
            raise SystemExit("VERIFICATION FAILED: the device generator differs from tests/datagen.bench_series")
        fitted, fit_cpu_seconds = ora.compress_chunks_timed(host_ts, fit_sample_values, offsets, eb, cores,
                                                            repetitions=3)
        gpu_fitted = downloaded.take(np.nonzero(downloaded.chunk_index < n_fit * chunks_per_series)[0])
        if fitted.rows() != gpu_fitted.rows():
            raise SystemExit("VERIFICATION FAILED: GPU segments differ from the oracle's")
        fit_rates = summary(fit_cpu_seconds, n_fit * args.points)
        fit_cpu = {"points_per_s": fit_rates["median"], "min": fit_rates["min"], "max": fit_rates["max"],
                   "repetitions": fit_rates["repetitions"],
                   "segments_per_s": len(fitted) / statistics.median(fit_cpu_seconds), "cores": cores,
                   "kind": "port",
                   "sample": f"{n_fit} series x {args.points} points, chunks sharded over {cores} pinned threads"}
        verified = {"fit_segments": len(fitted), "fit_points": n_fit * args.points,
                    "grid_points": grid_points_verified, "generator_points": len(host_defined),
                    "series_blocks": args.series,
                    "how": "after the timed region: oracle fit of the sample series' exact bytes == GPU "
                           "segments (all columns, byte for byte); oracle grid of the sample == the device "
                           "columns the timed step wrote (bit for bit); device generator == host definition; "
                           "every series of the rank starts and ends its block of the output columns where it must"}
        PHASES["verify_fit_and_cpu_baseline_fit"] = time.perf_counter() - verify_started
        print(f"[bench] verify_fit_and_cpu_baseline_fit: {PHASES['verify_fit_and_cpu_baseline_fit']:.2f} s", file=sys.stderr, flush=True)
        if not args.no_host_path:
            with phase("host_path"):
                host_path_result = host_path(context, mdb, np, downloaded, args)
            with phase("host_path_fit"):
                host_path_result["fit"] = host_fit(context, mdb, np, ora, host_ts, fit_sample_values, offsets, eb,
                                                   gpu_fitted)
        del downloaded
        if not args.no_irregular:
            with phase("irregular_timestamps"):
                irregular_result = irregular_timestamps(context, mdb, np, args)

    if rank == 0:
        value = world * points_per_step * args.steps / elapsed
        fit_models_ms = fit_kernel_ms.get("k_fit_models", 0.0) + fit_kernel_ms.get("k_fit_models_split", 0.0)
        fit_gbps = 4.0 * fit_points / (fit_models_ms * 1e-3) / 1e9 if fit_models_ms > 0 else 0.0
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
                "device": info["name"],
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
            "fit": {
                "cpu_baseline": fit_cpu,
                "points_per_s": fit_points / fit_seconds if fit_seconds > 0 else None,
                "segments_per_s": n_segments / fit_seconds if fit_seconds > 0 else None,
                "seconds": fit_seconds,
                "roofline": {"bound": "hbm", "kernel": "k_fit_models", "achieved": fit_gbps,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fit_gbps / HBM_PEAK_GBPS,
                             "kernel_ms": fit_models_ms,
                             "algorithmic_bytes_per_launch": 4.0 * fit_points,
                             "note": "4 B/point read (regular timestamps are synthesised, not loaded); the "
                                     "greedy fit is a sequential dependency per chunk, so this kernel is "
                                     "latency/issue-bound, far from the HBM roofline by nature"},
                "kernels_ms": fit_kernel_ms,
                "note": "PMC-Mean/Swing/MacaqueV fit of this rank's series on the GPU (setup, not "
                        "in the timed region), regular timestamps synthesised on the fly",
            },
        }
        os.write(result_fd, (json.dumps(result) + "\n").encode())

    for part in parts:
        part.free()
    context.dev_free(out_ts)
    context.dev_free(out_val)
    context.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
