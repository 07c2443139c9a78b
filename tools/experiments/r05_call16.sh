set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05v
python3 -m pytest tests/test_gpu_ntt.py -x -q -m gpu > gpurun_out/r05v/pytest.txt 2>&1; tail -2 gpurun_out/r05v/pytest.txt
for rep in 1 2; do
for v in 3 0 2 4; do
  L=""; [ $v != 3 ] && L=$GRAFT_REPO_ROOT/tools/experiments/lib_c$v/libSEAL_HEVM.so
  echo "pair stages in forward pass C = $v"
  DACAPO_AMD_LIB=$L python3 tools/legs/ntt_full_check.py 4096 20 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['forward'], d['round_trip'], 'fwd', d['fwd_full_us'], 'inv', d['inv_full_us'], 'two-phase fwd', d['fwd_two_phase_us'])"
  DACAPO_AMD_LIB=$L python3 tools/legs/ntt_full_check.py 1024 20 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['forward'], '1024: fwd', d['fwd_full_us'], 'inv', d['inv_full_us'])"
done; done > gpurun_out/r05v/c_pairs.txt 2>&1
cat gpurun_out/r05v/c_pairs.txt
