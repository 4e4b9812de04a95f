import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import modelardb_rs_amd as mdb
ctx = mdb.Context(0)
for size in (1 << 20, 64 << 20, 1300 << 20):
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); p = ctx.dev_alloc(size); t1 = time.perf_counter(); ctx.dev_free(p); t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    print(size >> 20, "MB: alloc / free ms", [(round(a, 3), round(f, 3)) for a, f in ts])
