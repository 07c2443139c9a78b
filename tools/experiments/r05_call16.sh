#!/bin/bash
# round 5, call 16: lazy sums through the group-walking inner-product kernel: parity, then config 4 (modes 1 / 2 / off)
mkdir -p gpurun_out/r05q
timeout 1500 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_config4_geometry.py -q -m gpu -k "lazy" > gpurun_out/r05q/pytest3.txt 2>&1
tail -8 gpurun_out/r05q/pytest3.txt
for lz in 1 2 0 1; do
  timeout 900 python tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 --opt hyb_lazy_sum=$lz > gpurun_out/r05q/c4_mode$lz.txt 2> gpurun_out/r05q/c4_mode$lz.err
  echo "mode $lz: $(tail -1 gpurun_out/r05q/c4_mode$lz.txt | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["run_s"], r["rms_vs_torch"], r.get("lazy_sums"))')"
done
