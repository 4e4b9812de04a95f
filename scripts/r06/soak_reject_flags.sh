cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
tail -3 scripts/gpu_soaks_r06.sh | grep -v "^#" > /tmp/two.sh; bash /tmp/two.sh
grep -n "MDB_FIT_PIECE_POINTS=64 MDB_GRID_MV_INDEX=0" scripts/gpu_soaks_r06.sh | cut -d: -f2- > /tmp/four.sh; bash /tmp/four.sh
