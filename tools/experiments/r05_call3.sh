set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_ntt.py tests/test_gpu_hevm.py tests/test_gpu_suite.py -x -q -m gpu > gpurun_out/r05c/pytest.txt 2>&1
tail -3 gpurun_out/r05c/pytest.txt
python3 tools/legs/per_op_sweep.py 30 cols_pairs=0 tiny_tile_wgs=512 cols_pairs=0,tiny_tile_wgs=512 > gpurun_out/r05c/per_op_sweep.txt 2>&1
python3 tools/legs/lowering_sweep.py 6 cols_pairs=0 ks_items_fast=0 cols_pairs=0,ks_items_fast=0,tiny_tile_wgs=512 > gpurun_out/r05c/lowering_sweep.txt 2>&1
cat gpurun_out/r05c/per_op_sweep.txt gpurun_out/r05c/lowering_sweep.txt
