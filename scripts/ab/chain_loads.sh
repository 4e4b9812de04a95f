for l in 8 16 32; do echo "loads $l"; MDB_AGG_CHAIN_LOADS=$l timeout 300 python scripts/profile_lossless_agg.py 2>&1 | tail -3 | grep -o "k_agg_mv_chains': [0-9.]*" | tr '\n' ' '; echo; MDB_AGG_CHAIN_LOADS=$l timeout 600 python - <<'PY'
import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np
import modelardb_rs_amd as mdb
ctx = mdb.Context(0)
eb = mdb.error_bound("lossless")
mask = mdb.MDB_AGG_SUM | mdb.MDB_AGG_COUNT
for streams in (1000, 100000):
    n = streams * 50000
    values = ctx.dev_alloc(4 * n); ctx.synth_values_dev(values, 0, streams, 50000)
    offsets = ctx.upload_array(np.arange(0, n + 50000, 50000, dtype=np.uint64))
    dev = ctx.compress_chunks_dev(0, values, offsets, streams, eb, 0, 1000, 0)
    ctx.dev_free(values)
    ctx.agg_batch_dev(dev, mask)
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(3): ctx.agg_batch_dev(dev, mask)
    print(streams, {k: round(v[1] / v[0], 3) for k, v in ctx.profile().items() if "chains" in k})
    ctx.profile_enable(False); dev.free()
PY
done
