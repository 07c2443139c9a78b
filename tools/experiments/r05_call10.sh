set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_prime_widths.py tests/test_gpu_config4_geometry.py -x -q -m gpu > gpurun_out/r05k/pytest.txt 2>&1
tail -5 gpurun_out/r05k/pytest.txt
python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 > gpurun_out/r05k/config4_9_8.txt 2>&1; tail -1 gpurun_out/r05k/config4_9_8.txt | cut -c1-500
python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 8 7 > gpurun_out/r05k/config4_8_7.txt 2>&1; tail -1 gpurun_out/r05k/config4_8_7.txt | cut -c1-500
python3 tools/legs/hybrid_ks_bench.py 17 40 9 8 10 0 > gpurun_out/r05k/hop_9_8.txt 2>&1
python3 tools/legs/hybrid_ks_bench.py 17 39 8 7 10 0 > gpurun_out/r05k/hop_8_7.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bt -- python3 $GRAFT_REPO_ROOT/tools/boot_demo.py 17 5 1 14 9 8 > $GRAFT_REPO_ROOT/gpurun_out/r05k/boot_9_8.txt 2>/dev/null
cp $(ls /tmp/bt/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/r05k/boot_9_8_kernel_stats.csv
