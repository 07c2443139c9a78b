set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05ac
python3 tools/legs/per_op_sweep.py 30 wide_tile_wgs=0 wide_tile_wgs=1024 wide_tile_wgs=4096 > gpurun_out/r05ac/per_op.txt 2>&1
python3 tools/legs/lowering_sweep.py 6 wide_tile_wgs=0 wide_tile_wgs=512 wide_tile_wgs=2048 wide_tile_wgs=8192 > gpurun_out/r05ac/low.txt 2>&1
cat gpurun_out/r05ac/per_op.txt gpurun_out/r05ac/low.txt gpurun_out/r05ac/hop.txt
