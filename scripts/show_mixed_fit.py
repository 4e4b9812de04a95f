import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("lossless","relative_1_percent"):
    f=d["mixed_models"][k]["fit"]; print(k, round(f["ms"],2), {a:round(b,2) for a,b in f["kernels_ms"].items() if b>0.5})
print("headline", d["value"], d["ms_per_step"], "fit", d["fit"]["seconds"])
