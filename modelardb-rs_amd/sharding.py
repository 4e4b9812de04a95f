"""Multi-GPU sharding of the hot path.

Series are independent units: compression state never crosses a series
(crates/modelardb_compression/src/compression.rs:224-263) and grid/sum/len are pure per-segment
functions (models/mod.rs:98-251). So rank r owns a contiguous block of series, fits and grids only
its own segments, and NO data-path collective exists. The single exchange step is the final merge
of the aggregate partials {f64 sum, i64 count, f32 min, f32 max}: 24 bytes per rank, latency-bound
on xGMI. It is done as an all-gather + local reduce in rank order so the f64 sum is reproducible
run to run (an all-reduce's ring order is not specified).
"""

import struct

import numpy as np

from . import _abi


def series_range(n_series, rank, world):
    """Contiguous block [first, last) of the series owned by `rank` (sizes differ by at most 1)."""
    base, extra = divmod(n_series, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def owner_of_series(series, n_series, world):
    base, extra = divmod(n_series, world)
    boundary = extra * (base + 1)
    if series < boundary:
        return series // (base + 1)
    return extra + (series - boundary) // max(base, 1)


def pack_state(state):
    """24 bytes: exactly the fields of mdb_agg_state, bit patterns preserved."""
    return struct.pack("<dqff", state.sum, state.count, state.min, state.max)


def unpack_state(data):
    total, count, mn, mx = struct.unpack("<dqff", bytes(data))
    return _abi.AggStateC(total, count, mn, mx)


def merge_states(states):
    """Fold partial states in the given order with the accumulators' own rules
    (crates/modelardb_storage/src/optimizer/model_simple_aggregates.rs:355, 398, 441, 501): the
    fold of the C ABI (mdb_agg_merge, csrc/mdb_comm.hip), the same one mdb_agg_all_reduce applies to
    what its all-gather returns."""
    import ctypes as C
    lib = _abi.load_hip_library()
    out = _abi.AggStateC.fresh()
    for state in states:
        part = _abi.AggStateC(state.sum, state.count, state.min, state.max)
        if lib.mdb_agg_merge(C.byref(out), C.byref(part)) != 0:
            raise RuntimeError(lib.mdb_last_error().decode())
    return out


def init_comm(context, dist):
    """Give `context` its RCCL communicator (mdb_comm_init): rank 0 makes the unique id through the
    C ABI and the process group that launched the ranks hands it round (any channel would do - a
    Rust host would use its own)."""
    from . import api
    rank, world = dist.get_rank(), dist.get_world_size()
    payload = [api.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(payload, src=0)
    context.comm_init(rank, world, payload[0])


def all_reduce_state(state, dist=None, device=None, context=None):
    """Merge every rank's partial aggregate state; every rank gets the same result.
    With `context` (a Context after init_comm) the exchange is the C ABI's mdb_agg_all_reduce: one
    ncclAllGather over RCCL / xGMI on the context's stream. Without, `dist` (torch.distributed) moves
    the bytes - gloo in the CPU tests - and the same fold is applied."""
    if context is not None:
        return context.agg_all_reduce(state)[0]
    import torch
    world = dist.get_world_size()
    payload = torch.frombuffer(bytearray(pack_state(state)), dtype=torch.uint8).clone()
    if device is not None:
        payload = payload.to(device)
    gathered = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)
    return merge_states(unpack_state(t.cpu().numpy().tobytes()) for t in gathered)
