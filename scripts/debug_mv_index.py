import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb
ctx = mdb.Context(0)
eb = mdb.error_bound("lossless")
per = 50_000
for streams in [int(x) for x in sys.argv[1:]]:
    n = streams * per
    values = ctx.dev_alloc(4 * n)
    ctx.synth_values_dev(values, 0, streams, per)
    offsets = np.arange(0, n + per, per, dtype=np.uint64)
    offsets_dev = ctx.upload_array(offsets)
    dev = ctx.compress_chunks_dev(0, values, offsets_dev, streams, eb, 0, 1000, 0)
    ctx.sync(); print(streams, "fitted", len(dev), flush=True)
    ctx.dev_free(values)
    total = ctx.grid_count_dev(dev); print("count", total, flush=True)
    out_ts, out_val = ctx.dev_alloc(8 * total), ctx.dev_alloc(4 * total)
    for k in range(2):
        t = time.perf_counter()
        ctx.grid_batch_dev(dev, out_ts, out_val, total); ctx.sync()
        print("grid", k, time.perf_counter() - t, flush=True)
    # spot checks far into the output: series s, value j is bench_series(s, j)
    import datagen
    for s_index in (0, streams // 2, streams - 1):
        got = ctx.download_array(out_val, per, np.float32, offset_elements=s_index * per)
        assert np.array_equal(got.view(np.uint32), datagen.bench_series(s_index, per).view(np.uint32)), s_index
    print("values checked", flush=True)
    mask = mdb.MDB_AGG_SUM | mdb.MDB_AGG_COUNT
    st = ctx.agg_batch_dev(dev, mask); print("agg", st.count, flush=True)
    for p in (out_ts, out_val, offsets_dev): ctx.dev_free(p)
    dev.free()
