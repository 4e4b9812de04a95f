import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("lossless","relative_1_percent"):
    m=d["mixed_models"][k]
    for name in ("aggregates","aggregates_between_quartiles"):
        a=m[name]; print(k, name, round(a["ms"],3), a["kernels_ms"])
