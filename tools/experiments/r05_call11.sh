set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05n
python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_hevm.py tests/test_gpu_suite.py tests/test_gpu_prime_widths.py -x -q -m gpu > gpurun_out/r05n/pytest.txt 2>&1
tail -5 gpurun_out/r05n/pytest.txt
for rep in 1 2; do
python3 tools/legs/lowering_sweep.py 6 > gpurun_out/r05n/low_new_$rep.txt 2>&1
DACAPO_AMD_LIB=$GRAFT_REPO_ROOT/tools/experiments/lib_prev/libSEAL_HEVM.so python3 tools/legs/lowering_sweep.py 6 > gpurun_out/r05n/low_prev_$rep.txt 2>&1
python3 tools/legs/per_op_sweep.py 30 > gpurun_out/r05n/op_new_$rep.txt 2>&1
DACAPO_AMD_LIB=$GRAFT_REPO_ROOT/tools/experiments/lib_prev/libSEAL_HEVM.so python3 tools/legs/per_op_sweep.py 30 > gpurun_out/r05n/op_prev_$rep.txt 2>&1
done
grep -H defaults gpurun_out/r05n/low_*.txt gpurun_out/r05n/op_*.txt
