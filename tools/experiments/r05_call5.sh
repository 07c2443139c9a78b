set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
python3 -m pytest tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r05e/pytest_rccl.txt 2>&1
tail -15 gpurun_out/r05e/pytest_rccl.txt
python3 -m pytest tests/test_gpu_hevm.py tests/test_gpu_seal_io.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r05e/pytest_hevm.txt 2>&1
tail -5 gpurun_out/r05e/pytest_hevm.txt
