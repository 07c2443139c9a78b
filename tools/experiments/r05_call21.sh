#!/bin/bash
# round 5, call 21: ROWS phases prime by prime (option rows_prime_major): parity at N = 2^17, then config 4 on / off on one box
mkdir -p gpurun_out/r05q
timeout 1500 python -m pytest tests/test_gpu_config4_geometry.py tests/test_gpu_ntt.py -q -m gpu -x > gpurun_out/r05q/pytest6.txt 2>&1; tail -3 gpurun_out/r05q/pytest6.txt
run() { timeout 900 python tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 --opt hyb_lazy_sum=1 "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["run_s"], r["rms_vs_torch"])'; }
for o in "--opt rows_prime_major=16" "--opt rows_prime_major=0" "--opt rows_prime_major=16" "--opt rows_prime_major=0"; do echo "[$o] $(run $o)"; done | tee gpurun_out/r05q/c4_prime_major.txt
python tools/legs/hybrid_ks_bench.py 17 40 9 8 10 0 --opt rows_prime_major=16 2>/dev/null | tail -1 | cut -c1-600
python tools/legs/hybrid_ks_bench.py 17 40 9 8 10 0 --opt rows_prime_major=0 2>/dev/null | tail -1 | cut -c1-600
