#!/bin/bash
mkdir -p gpurun_out/r05q
for m in 2 3; do timeout 900 python tools/experiments/r05_rows_mac_probe.py $m parity 2>&1 | tail -3; done
for m in 0 2 3 0 2 3; do timeout 600 python tools/experiments/r05_rows_mac_probe.py $m 2>&1 | tail -1; done
