#!/bin/bash
# SQ counters of the fit's model kernel at the headline shape (1 000 series x 10 M points, relative 1 %): four passes
# of rocprofv3 --pmc (each with --kernel-trace only), per wave of the last launch; writes gpurun_out/pmc_fit_models.json
# (copied to profiles/ by hand: bench.py's fit roofline reads valu_per_point from it while the source hash matches).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
KERNEL=${1:-k_fit_models_lean}
cd $ROOT
PMC_TAG=fitmodels scripts/r04/pmc_kernel4.sh $KERNEL scripts/profile_fit.py --series 1000 --points 10000000 > $OUT/pmc_fit_models.txt 2>&1
cat $OUT/pmc_fit_models.txt
python3 - <<PY
import ast, hashlib, json, re
text = open("$OUT/pmc_fit_models.txt").read()
counters, waves = {}, None
for line in text.splitlines():
    m = re.match(r"(\S+) pass \w: waves ([\d.]+) launches \d+ (\{.*\})", line)
    if m:
        waves = float(m.group(2)); counters.update(ast.literal_eval(m.group(3)))
digest = hashlib.sha256()
for name in ("mdb_fit.hip", "mdb_segment_dev.hpp", "mdb_common.hpp"):
    digest.update(open("$ROOT/modelardb-rs_amd/csrc/" + name, "rb").read())
# (rotation: a few waves fewer than groups of 64 chunks take the groups through - per GROUP, not per wave of the launch)
groups = -(-(1000 * -(-10000000 // 65536)) // 64)
points = 65536.0 * groups / waves
out = {"kernel": "$KERNEL", "series": 1000, "points": 10000000, "waves": waves, "source_hash": digest.hexdigest()[:16],
       "per_wave": counters,
       "valu_per_point": counters.get("SQ_INSTS_VALU", 0) / points, "salu_per_point": counters.get("SQ_INSTS_SALU", 0) / points,
       "branch_per_point": counters.get("SQ_INSTS_BRANCH", 0) / points,
       "groups_of_64_chunks": groups,
       "note": "per wave of the last launch x waves / groups of 64 chunks / 65 536 (the points of a lane's chunk; one step of a wave = one point of each of a group's 64 chunks)"}
json.dump(out, open("$OUT/pmc_fit_models.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_wave"}))
PY
