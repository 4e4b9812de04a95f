"""TEST INFRASTRUCTURE: bench.orchestrate() - the rank protocol of bench.py - over gloo on CPU with a canned
workload (no kernels, no library): every rank "grids" a fixed number of points per step, the aggregate partials
are merged by an all-gather and a fold in rank order as mdb_agg_all_reduce does, rank 0 alone has a tail.
Started by tests/test_bench_cpu.py under torch.distributed.run with WORLD_SIZE = 2. FAIL_ON=<rank>:<where>
makes that rank raise in `step` or in `tail`."""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class CannedWorkload:
    POINTS_PER_STEP = 1_000_000

    def __init__(self, args, rank, local_rank, world, dist):
        self.args, self.rank, self.world, self.dist = args, rank, world, dist
        self.fail_rank, _, self.fail_where = os.environ.get("FAIL_ON", "-1:").partition(":")
        self.closed = False

    def _maybe_fail(self, where):
        if int(self.fail_rank) == self.rank and self.fail_where == where:
            raise RuntimeError(f"rank {self.rank} fails in {where} (asked to)")

    def build(self):
        pass

    def step(self):
        self._maybe_fail("step")
        time.sleep(0.001 * (1 + self.rank))
        return self.POINTS_PER_STEP

    def sync(self):
        pass

    def report(self, elapsed, per_rank_seconds):
        import torch
        # the merge of the partial aggregate states: every rank (a collective), folded in rank order
        mine = torch.tensor([float(self.rank + 1), float(self.POINTS_PER_STEP)], dtype=torch.float64)
        states = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(states, mine)
        count = int(sum(state[1].item() for state in states))
        if self.rank != 0:
            return None
        time.sleep(0.3)  # rank 0's tail: the others wait at the meeting point
        self._maybe_fail("tail")
        return {"metric": "gridded values/sec", "value": self.world * self.POINTS_PER_STEP * self.args.steps / elapsed,
                "n_gpus": self.world, "steps": self.args.steps, "warmup": self.args.warmup,
                "ms_per_step": 1e3 * elapsed / self.args.steps, "rccl_ranks_seen": len(states),
                "ms_per_step_per_rank": {"min": 1e3 * min(per_rank_seconds) / self.args.steps,
                                         "max": 1e3 * max(per_rank_seconds) / self.args.steps},
                "aggregates": {"result": {"count": count}}, "data": "canned"}

    def close(self):
        self.closed = True
        print(f"rank {self.rank} closed", file=sys.stderr, flush=True)


if __name__ == "__main__":
    arguments = types.SimpleNamespace(gpus=int(os.environ["WORLD_SIZE"]), steps=3, warmup=1, collective_timeout=20.0,
                                      detail_file=os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_detail_stub.json"))
    sys.exit(bench.orchestrate(arguments, CannedWorkload, backend="gloo", result_fd=1))
