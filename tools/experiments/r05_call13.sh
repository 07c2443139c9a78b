#!/bin/bash
# round 5, call 13: lazy sums (option hyb_lazy_sum) -- parity test, then config 4 with and without
mkdir -p gpurun_out/r05q
timeout 900 python -m pytest tests/test_gpu_hybrid.py -x -q -m gpu -k "lazy or share or bit_identical" > gpurun_out/r05q/pytest.txt 2>&1
tail -15 gpurun_out/r05q/pytest.txt
for lz in 1 0; do
  timeout 900 python tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 --opt hyb_lazy_sum=$lz --opt trace=1 > gpurun_out/r05q/config4_lazy$lz.txt 2> gpurun_out/r05q/config4_lazy$lz.err
  tail -c 1500 gpurun_out/r05q/config4_lazy$lz.txt
  grep -E "rotsum|plan:   rot |plan: [0-9]" gpurun_out/r05q/config4_lazy$lz.err | head
done
