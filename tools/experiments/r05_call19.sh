#!/bin/bash
# round 5, call 19: lazy sums on the generic-width build (test + config 4 on the mixed chain), then the bench line again
mkdir -p gpurun_out/r05q gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_prime_widths.py -q -m gpu -k "lazy" > gpurun_out/r05q/pytest5.txt 2>&1; tail -3 gpurun_out/r05q/pytest5.txt
for lz in 1 0; do
  timeout 900 python tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14r51 9 8 mixed_app --opt hyb_lazy_sum=$lz > gpurun_out/r05q/c4_mixed_lazy$lz.txt 2> gpurun_out/r05q/c4_mixed_lazy$lz.err
  echo "mixed lazy $lz: $(tail -1 gpurun_out/r05q/c4_mixed_lazy$lz.txt | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["run_s"], r["rms_vs_torch"], r.get("lazy_sums"))')"
done
python bench.py > gpurun_out/r05/r05_bench.json 2> gpurun_out/r05/bench.err
tail -c 300 gpurun_out/r05/r05_bench.json
