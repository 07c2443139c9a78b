set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05r
NL=$GRAFT_REPO_ROOT/tools/experiments/lib_nonop/libSEAL_HEVM.so
DACAPO_AMD_LIB=$NL python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_hevm.py tests/test_gpu_suite.py -x -q -m gpu > gpurun_out/r05r/pytest.txt 2>&1
tail -3 gpurun_out/r05r/pytest.txt
for rep in 1 2; do
DACAPO_AMD_LIB=$NL python3 tools/legs/lowering_sweep.py 6 > gpurun_out/r05r/low_nonop_$rep.txt 2>&1
python3 tools/legs/lowering_sweep.py 6 > gpurun_out/r05r/low_base_$rep.txt 2>&1
DACAPO_AMD_LIB=$NL python3 tools/legs/per_op_sweep.py 30 > gpurun_out/r05r/op_nonop_$rep.txt 2>&1
python3 tools/legs/per_op_sweep.py 30 > gpurun_out/r05r/op_base_$rep.txt 2>&1
done
grep -H defaults gpurun_out/r05r/low_*.txt gpurun_out/r05r/op_*.txt
