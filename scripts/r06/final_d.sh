#!/bin/bash
# Round 6, after the last change to mdb_fit.hip: the fit kernel's SQ counters again (bench.py's valu_issue is tied to the
# sources' hash), then - second call, with the new profiles/pmc_fit_models.json in the tree - final_c.sh.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
bash scripts/r05/pmc_fit.sh > $OUT/pmc_fit.log 2>&1; echo "pmc fit rc=$?"
cp $ROOT/gpurun_out/pmc_fit_models.json $ROOT/gpurun_out/pmc_fit_models.txt $OUT/ 2>/dev/null
timeout 600 python3 scripts/r04/mixed_fit.py 1e9 lossless,rel1 > $OUT/mixed_fit.log 2>&1; echo "mixed fit rc=$?"
tail -2 $OUT/mixed_fit.log | cut -c 1-300
tail -1 $OUT/pmc_fit.log | cut -c 1-400
