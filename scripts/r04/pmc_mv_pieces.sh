#!/bin/bash
# SQ counters of the piece decoders of MacaqueV streams on the mixed series (scripts/r04/mixed_grid.py), per wave of the
# kernels' last launch: k_grid_mv_pieces (a wave = 64 pieces of at most 64 values: vector instructions per value step
# = SQ_INSTS_VALU / 64 where the pieces are full) and k_agg_mv_pieces; written to gpurun_out/r04/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
PMC_TAG=mvgrid scripts/r04/pmc_kernel4.sh k_grid_mv_pieces scripts/r04/mixed_grid.py > $OUT/pmc_mv_pieces_sq_counters.txt 2>&1
PMC_TAG=mvagg scripts/r04/pmc_kernel4.sh k_agg_mv_pieces scripts/r04/mixed_grid.py >> $OUT/pmc_mv_pieces_sq_counters.txt 2>&1
cat $OUT/pmc_mv_pieces_sq_counters.txt
