import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("lossless","relative_1_percent"):
    f=d["mixed_models"][k]["fit"]; print(k, round(f["ms"],2), {a:round(b,2) for a,b in f["kernels_ms"].items() if b>0.5})
print("headline", d["value"], d["ms_per_step"], "fit", d["fit"]["seconds"])
for k in ("lossless","relative_1_percent"):
    h=d["mixed_models"][k].get("host_path")
    if h: print(k, "host path", h)
    g=d["mixed_models"][k]["grid"]; print(k, "grid ms", round(g["ms"],2), g["kernels_ms"])
print("sum accumulator (swing sample)", d.get("host_path",{}).get("sum_accumulator_batch_8192"))
