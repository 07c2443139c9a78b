set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05p
python3 -m pytest tests/ -x -q -m gpu > gpurun_out/r05p/pytest_gpu.txt 2>&1
tail -8 gpurun_out/r05p/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05p/smoke.txt 2>&1; tail -3 gpurun_out/r05p/smoke.txt
