set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05q
python3 tools/legs/lowering_sweep.py 5 --only b13 ks_big_tiles=2048 ks_big_tiles=8192 ks_big_tiles=16384 small_tile_wgs=2500 small_tile_wgs=10000 ks_merge_special_min_wgs=1024 ks_merge_special_min_wgs=4096 ks_merge_lift_min_wgs=512 ks_merge_lift_min_wgs=4096 sum_pair_min_wgs=512 sum_group_min_wgs=1024 ntt_full_min_limbs=100000 > gpurun_out/r05q/b13_shapes.txt 2>&1
python3 tools/legs/lowering_sweep.py 5 --only b13 --new-vm max_batch=32 max_batch=48 max_batch=96 max_batch=128 plan_aux_min_cost=1 plan_aux_min_cost=8 chain_fusion=0 > gpurun_out/r05q/b13_vm.txt 2>&1
cat gpurun_out/r05q/b13_shapes.txt gpurun_out/r05q/b13_vm.txt
