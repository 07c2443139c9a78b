#!/bin/bash
# round 5, call 17: the whole GPU suite + smoke on the lazy-sums build
mkdir -p gpurun_out/r05p
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r05p/pytest_gpu.txt 2>&1
tail -6 gpurun_out/r05p/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05p/smoke.txt 2>&1; tail -3 gpurun_out/r05p/smoke.txt
sha256sum dacapo_amd/lib/*.so
