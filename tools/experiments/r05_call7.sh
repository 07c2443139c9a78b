set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
python3 -m pytest tests/test_gpu_hevm.py -x -q -m gpu -k "duplicate_rotations or explicit_dag or bounded_rotation" > gpurun_out/r05g/pytest_adv.txt 2>&1
tail -30 gpurun_out/r05g/pytest_adv.txt
python3 -m pytest tests/test_gpu_ntt.py tests/test_gpu_boot.py -x -q -m gpu > gpurun_out/r05g/pytest_ntt.txt 2>&1
tail -3 gpurun_out/r05g/pytest_ntt.txt
for n in 640 768 900 1024 1300 1536 2048; do python3 tools/legs/ntt_full_check.py $n 20; done > gpurun_out/r05g/ntt_full_check.txt 2>&1
python3 tools/legs/lowering_sweep.py 6 ntt_full_inv_pairs=0 > gpurun_out/r05g/lowering.txt 2>&1; cat gpurun_out/r05g/lowering.txt
