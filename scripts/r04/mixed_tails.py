import os, sys
import numpy as np
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"tests")]
import modelardb_rs_amd as mdb, datagen
ctx=mdb.Context(0)
points=1_000_000
for s in (0,1):
    values=datagen.mixed_series(points, 1000+s, (1.0,1.05) if s%2 else None)[1]
    ts=1000*np.arange(points,dtype=np.int64)
    offsets=np.arange(0,points+65536,65536,dtype=np.uint64); offsets[-1]=points
    for name,eb in (("rel1",mdb.error_bound("relative",1.0)),("lossless",mdb.error_bound("lossless"))):
        seg=ctx.compress_chunks(ts,values,offsets,eb)
        res=seg.residuals.to_bytes_list()
        counts=np.array([r[-1] if len(r) else 0 for r in res])
        types=np.bincount(seg.model_type_id,minlength=3)
        lens=(seg.end_time-seg.start_time)//1000+1
        mv=lens[seg.model_type_id==2].sum()
        w=counts>0
        print(f"series {s} {name}: {len(seg)} segments types {types}, tails {w.sum()} values {counts.sum()} ({counts.sum()/points:.3f} of points) mean {counts[w].mean():.1f} p50 {np.median(counts[w])} p90 {np.percentile(counts[w],90)}; MacaqueV model points {mv/points:.3f}; res bytes {seg.residuals.lengths().sum()/max(1,counts.sum()):.2f} B/value")
