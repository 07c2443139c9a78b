#!/bin/bash
# round 5, call 14: lazy sums -- parity tests (small ring, config-4 geometry)
mkdir -p gpurun_out/r05q
timeout 1500 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_config4_geometry.py -q -m gpu -k "lazy" > gpurun_out/r05q/pytest2.txt 2>&1
tail -40 gpurun_out/r05q/pytest2.txt
