#!/bin/bash
# round 5, call 20: config 4 with lazy sums under a few plan / launch-shape options (same box)
mkdir -p gpurun_out/r05q
run() { timeout 900 python tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 --opt hyb_lazy_sum=1 "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["run_s"], r["rms_vs_torch"])'; }
for o in "" "--opt max_batch=32" "--opt max_batch=128" "--opt max_batch=256" "--opt plan_lanes=1" "--opt hyb_fuse=1" "--opt sum_group_min_wgs=-1" "--opt plan_graph=2" ""; do
  echo "[$o] $(run $o)"
done | tee gpurun_out/r05q/c4_sweep.txt
