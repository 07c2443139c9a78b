set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05t
python3 tools/legs/per_op_sweep.py 30 ks_mac_tiny_wgs=0 ks_mac_tiny_wgs=1000 ks_mac_tiny_wgs=4000 ks_mac_tiny_wgs=100000 > gpurun_out/r05t/mac_tiny.txt 2>&1
python3 tools/legs/lowering_sweep.py 6 ks_mac_tiny_wgs=0 ks_mac_tiny_wgs=1000 ks_mac_tiny_wgs=4000 ks_mac_tiny_wgs=8000 ks_mac_tiny_wgs=100000 > gpurun_out/r05t/mac_tiny_low.txt 2>&1
cat gpurun_out/r05t/mac_tiny.txt gpurun_out/r05t/mac_tiny_low.txt
