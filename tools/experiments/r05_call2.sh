set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05b
bash $R/tools/collect_run_budget.sh r05b b13 > $R/gpurun_out/r05b/b13.log 2>&1
for o in tiny_tile_wgs=512 tiny_tile_wgs=1000 tiny_tile_wgs=2000 tiny_tile_wgs=4000; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$o -- python3 $R/tools/per_op_only.py 20 --only rotate_hop --opt $o > /dev/null 2>/tmp/kt.err
  echo "== rotate_hop, $o"; python3 $R/tools/summarize_trace.py $(ls /tmp/kt_$o/*/*kernel_trace.csv | head -1) | grep -v rocclr | head -12
done > $R/gpurun_out/r05b/rotate_tiny_kernels.txt 2>&1
cd $R
python3 tools/legs/lowering_sweep.py 6 tiny_tile_wgs=1000 tiny_tile_wgs=2000 tiny_tile_wgs=4000 tiny_tile_wgs=8000 > gpurun_out/r05b/lowering_sweep.txt 2>&1
python3 tools/legs/per_op_sweep.py 30 tiny_tile_wgs=2000 tiny_tile_wgs=4000 tiny_tile_wgs=8000 > gpurun_out/r05b/per_op_sweep.txt 2>&1
python3 -m pytest tests/test_gpu_hevm.py -x -q -m gpu > gpurun_out/r05b/pytest_hevm.txt 2>&1
tail -5 gpurun_out/r05b/pytest_hevm.txt; cat gpurun_out/r05b/lowering_sweep.txt gpurun_out/r05b/per_op_sweep.txt
