#!/bin/bash
# round 5, call 23: twiddle pairs in the generic forward ROWS phase (option rows_pairs): parity, then config 4 and the hop on / off
mkdir -p gpurun_out/r05q
timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_config4_geometry.py tests/test_gpu_ops.py -q -m gpu -x > gpurun_out/r05q/pytest7.txt 2>&1; tail -3 gpurun_out/r05q/pytest7.txt
run() { timeout 900 python tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 --opt hyb_lazy_sum=1 "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["run_s"], r["rms_vs_torch"])'; }
for o in "--opt rows_pairs=1" "--opt rows_pairs=0" "--opt rows_pairs=1" "--opt rows_pairs=0"; do echo "[$o] $(run $o)"; done | tee gpurun_out/r05q/c4_rows_pairs.txt
for v in 1 0; do python tools/legs/hybrid_ks_bench.py 17 40 9 8 10 0 --opt rows_pairs=$v 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print([(l["level"], l["hop_us"]) for l in d["levels"]])'; done
