#!/bin/bash
# round 5, call 18: lazy sums also for the last hop of a composed rotation: parity, then config 4 under 49 / 96 / 286 keys, lazy on / off
mkdir -p gpurun_out/r05q
timeout 1500 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_config4_geometry.py tests/test_gpu_config4.py -q -m gpu > gpurun_out/r05q/pytest4.txt 2>&1
tail -5 gpurun_out/r05q/pytest4.txt
for keys in 49 96 1; do for lz in 1 0; do
  timeout 900 python tools/legs/resnet_real_boot.py $keys resnet20_nt16 17 1 b14 9 8 --opt hyb_lazy_sum=$lz > gpurun_out/r05q/c4_k${keys}_lazy$lz.txt 2> gpurun_out/r05q/c4_k${keys}_lazy$lz.err
  echo "keys $keys lazy $lz: $(tail -1 gpurun_out/r05q/c4_k${keys}_lazy$lz.txt | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["run_s"], r["rms_vs_torch"], r["key_switches"], r.get("lazy_sums"))')"
done; done
