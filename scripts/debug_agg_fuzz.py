#!/usr/bin/env python3
"""Replays tests/test_gpu_agg.py::test_fuzzed_segments... and, for every batch on which the oracle
and the GPU disagree about acceptance, shows per row who accepts what. Development tool."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases, oracle_lib as ora, modelardb_rs_amd as mdb  # noqa: E402

hip = mdb.Context(0)
rng = np.random.default_rng(231)
pool = []
for eb_name in ("lossless", "rel5"):
    pool += cases.edge_case_batch(cases.error_bounds()[eb_name]).rows()
    for irregular in (False, True):
        pool += cases.mixed_batch(cases.error_bounds()[eb_name], irregular, seed=232, length=3000)[2].rows()

def accepts(call):
    try:
        call()
        return True
    except (ora.OracleError, mdb.HipError) as e:
        return str(e)[:60]

for trial in range(300):
    rows = []
    for _ in range(int(rng.integers(1, 6))):
        row = list(pool[int(rng.integers(0, len(pool)))])
        if rng.random() < 0.35:
            field = int(rng.choice([0, 1, 2, 3, 6, 7]))
            if field == 0:
                row[0] = int(rng.integers(0, 4))
            elif field in (1, 2):
                row[field] = int(row[field] + rng.integers(-500, 500))
            else:
                payload = bytearray(row[field])
                action = rng.integers(0, 3)
                if action == 0 and payload:
                    payload = payload[: int(rng.integers(0, len(payload)))]
                elif action == 1 and payload:
                    payload[int(rng.integers(0, len(payload)))] ^= 1 << int(rng.integers(0, 8))
                else:
                    payload = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 20)), dtype=np.uint8).tobytes())
                row[field] = bytes(payload)
        rows.append(tuple(row))
    for row in rows:
        b = mdb.SegmentBatch.from_rows([row])
        oa = accepts(lambda: ora.agg_batch(b, 15))
        if oa is True and ora.agg_batch(b, 15).count > 200000:
            continue
        og = accepts(lambda: ora.grid_batch(b))
        ha = accepts(lambda: hip.agg_batch(b, 15))
        hg = accepts(lambda: hip.grid_batch(b, cap=200000))
        if (oa is True and og is True) and ha is not True:
            print(trial, [x if not isinstance(x, bytes) else x[:16].hex() + f"({len(x)})" for x in row[:8]],
                  "| oracle agg", oa, "grid", og, "| hip agg", ha, "grid", hg, flush=True)
