set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
python3 -m pytest tests/test_gpu_ntt.py tests/test_gpu_hybrid.py tests/test_gpu_ops.py tests/test_gpu_prime_widths.py -x -q -m gpu > gpurun_out/r05i/pytest.txt 2>&1
tail -5 gpurun_out/r05i/pytest.txt
for o in cols_pairs=1 cols_pairs=0; do
python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 --opt $o > gpurun_out/r05i/config4_9_8_$o.txt 2>&1; tail -1 gpurun_out/r05i/config4_9_8_$o.txt | cut -c1-500
python3 tools/legs/hybrid_ks_bench.py 17 40 9 8 10 0 --opt $o > gpurun_out/r05i/hop_9_8_$o.txt 2>&1
done
python3 tools/legs/lowering_sweep.py 6 cols_pairs=0 > gpurun_out/r05i/lowering.txt 2>&1; cat gpurun_out/r05i/lowering.txt
