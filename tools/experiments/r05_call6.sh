set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
python3 -m pytest tests/test_gpu_ntt.py -x -q -m gpu > gpurun_out/r05f/pytest_ntt.txt 2>&1
tail -3 gpurun_out/r05f/pytest_ntt.txt
for n in 256 384 512 640 768 1024 1536 2048 4096; do python3 tools/legs/ntt_full_check.py $n 20; done > gpurun_out/r05f/ntt_full_check_pairs.txt 2>&1
for n in 1024 2048 4096; do python3 tools/legs/ntt_full_check.py $n 20 --opt ntt_full_inv_pairs=0; done > gpurun_out/r05f/ntt_full_check_words.txt 2>&1
cat gpurun_out/r05f/ntt_full_check_pairs.txt gpurun_out/r05f/ntt_full_check_words.txt
