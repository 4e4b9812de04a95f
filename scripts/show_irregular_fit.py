import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("regular","random_intervals","one_percent_gaps"):
    print(k, "fit_ms", round(d["irregular_timestamps"][k]["fit_ms"],2))
f=d["host_path"]["fit"] if "host_path" in d else None
if f: print("host fit", f["points_per_s"], [ (r["chunks"], r["gpu_ms"]) for r in f["fit_latency"]["rows"]])
