set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
python3 tools/legs/per_op_sweep.py 30 cols_pairs=0 > gpurun_out/r05d/per_op_sweep.txt 2>&1
bash tools/collect_run_budget.sh r05d b13 > gpurun_out/r05d/b13.log 2>&1
bash tools/collect_run_budget.sh r05d headline > gpurun_out/r05d/headline.log 2>&1
cat gpurun_out/r05d/per_op_sweep.txt
