set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05h
python3 -m pytest tests/test_gpu_config4_geometry.py tests/test_gpu_hybrid.py -x -q -m gpu > gpurun_out/r05h/pytest_c4geo.txt 2>&1
tail -15 gpurun_out/r05h/pytest_c4geo.txt
python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 > gpurun_out/r05h/config4_9_8.txt 2>&1; tail -2 gpurun_out/r05h/config4_9_8.txt | cut -c1-600
python3 tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 8 7 > gpurun_out/r05h/config4_8_7.txt 2>&1; tail -2 gpurun_out/r05h/config4_8_7.txt | cut -c1-600
python3 tools/legs/hybrid_ks_bench.py 17 40 9 8 10 0 > gpurun_out/r05h/hop_9_8.txt 2>&1; cat gpurun_out/r05h/hop_9_8.txt | cut -c1-400
python3 tools/legs/hybrid_ks_bench.py 17 39 8 7 10 0 > gpurun_out/r05h/hop_8_7.txt 2>&1; cat gpurun_out/r05h/hop_8_7.txt | cut -c1-400
