#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export MDB_HIP_LIBRARY=$PWD/scripts/ab/timing6_libmdb_hip.so MDB_FIT_TIMING_FILE=$PWD/gpurun_out/fit_waves_placement.csv
python3 scripts/profile_fit.py --series ${SERIES:-1000} --points 10000000 2>&1 | grep -E "fit timing|k_fit_models " | tail -4
python3 - <<'PY'
import csv, collections, os
rows = list(csv.DictReader(open(os.environ["MDB_FIT_TIMING_FILE"])))
per = collections.defaultdict(list)
t0 = min(int(r["start"]) for r in rows)
for r in rows:
    hw, xcc = int(r["hw_id"]), int(r["xcc_id"]) & 0xf
    # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)
    per[key].append((int(r["ticks"]) / 100.0, (int(r["start"]) - t0) / 100.0))
by_count = collections.defaultdict(list)
for key, waves in per.items():
    by_count[len(waves)].append(max(us for us, _ in waves) / 1e3)
print("SIMDs seen", len(per), "CUs", len({k[:4] for k in per}), "XCCs", len({k[0] for k in per}))
for count in sorted(by_count):
    v = by_count[count]
    print(f"SIMDs with {count} waves: {len(v)}; their slowest wave: mean {sum(v)/len(v):.1f} ms, min {min(v):.1f}, max {max(v):.1f}")
starts = sorted(s for w in per.values() for _, s in w)
print("wave start spread us: median", starts[len(starts)//2], "max", starts[-1])
cu = collections.Counter(k[:4] for k, w in per.items() for _ in w)
print("waves per CU:", sorted(collections.Counter(cu.values()).items()))
PY
