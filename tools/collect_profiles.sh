#!/bin/bash
# Everything under profiles/ for one round, collected on the GPU box in one gpurun call:
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r03'
# Outputs go to gpurun_out/<round>/ (merged back by gpurun); copy the summaries into profiles/ afterwards.
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# A. kernel trace + stats of the bench command (program directly after `--`)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lowerings --no-config4 > $OUT/bench_under_profiler.json 2> $OUT/kt.err
KT=$(ls $OUT/kt/*/*kernel_trace.csv | head -1)
cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/${R}_kernel_stats.csv
cd $ROOT
python tools/summarize_trace.py $KT > $OUT/${R}_by_kernel_and_grid.txt
python tools/summarize_trace.py $KT 4096 > $OUT/${R}_roofline_leg_launches.txt
python tools/timeline_gaps.py $KT > $OUT/${R}_timeline.txt
python tools/top_kernels.py $KT > $OUT/${R}_top_kernels.json
rm -rf $OUT/kt
# B. HBM traffic of the NTT phase kernels: FETCH_SIZE and WRITE_SIZE need separate passes
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pf -- python3 $ROOT/tools/ntt_only.py 15 4096 2 > /dev/null 2> $OUT/pf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pw -- python3 $ROOT/tools/ntt_only.py 15 4096 2 > /dev/null 2> $OUT/pw.err
cd $ROOT
python tools/collect_traffic.py $(ls $OUT/pf/*/*counter_collection.csv | head -1) $(ls $OUT/pw/*/*counter_collection.csv | head -1) > $OUT/${R}_ntt_hbm_traffic.json
rm -rf $OUT/pf $OUT/pw
# B2. the timed step's own kernels: durations, FETCH_SIZE / WRITE_SIZE per kernel, algorithmic bytes where the grid encodes (level, batch)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/hk -- python3 $ROOT/tools/headline_only.py 3 > /dev/null 2> $OUT/hk.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/hf -- python3 $ROOT/tools/headline_only.py 3 > /dev/null 2> $OUT/hf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/hw -- python3 $ROOT/tools/headline_only.py 3 > /dev/null 2> $OUT/hw.err
cd $ROOT
python tools/kernel_traffic.py $(ls $OUT/hk/*/*kernel_trace.csv | head -1) $(ls $OUT/hf/*/*counter_collection.csv | head -1) $(ls $OUT/hw/*/*counter_collection.csv | head -1) > $OUT/${R}_step_kernels.json
rm -rf $OUT/hk $OUT/hf $OUT/hw
# B3. config 4 (N = 2^17, 38 real bootstraps, grouped-digit keys): kernel-time table
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $ROOT/tools/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 8 7 > $OUT/${R}_config4_under_profiler.txt 2> $OUT/c4.err
cp $(ls $OUT/c4/*/*kernel_stats.csv | head -1) $OUT/${R}_config4_kernel_stats.csv
rm -rf $OUT/c4
cd $ROOT
# B4. one grouped-digit key switch at N = 2^17, top level: every kernel of the sequence on the byte roofline, with the two base conversions
#     on the matrix cores (default) and on the vector units (DACAPO_HYB_MFMA=0), and the matrix-core counters of the former
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hy -- python3 $ROOT/tools/hybrid_ks_bench.py 17 39 8 7 5 31 > $OUT/hy_hop.json 2> $OUT/hy.err
export DACAPO_HYB_MFMA=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hv -- python3 $ROOT/tools/hybrid_ks_bench.py 17 39 8 7 5 31 > $OUT/hv_hop.json 2> $OUT/hv.err
unset DACAPO_HYB_MFMA
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/hm -- python3 $ROOT/tools/hybrid_ks_bench.py 17 39 8 7 5 31 > /dev/null 2> $OUT/hm.err
cd $ROOT
{ echo "== base conversions on the matrix cores (default)"; python tools/hybrid_ks_summary.py $(ls $OUT/hy/*/*kernel_stats.csv | head -1) $OUT/hy_hop.json;
  echo; echo "== base conversions on the vector units (DACAPO_HYB_MFMA=0)"; python tools/hybrid_ks_summary.py $(ls $OUT/hv/*/*kernel_stats.csv | head -1) $OUT/hv_hop.json;
  echo; echo "== counters of the matrix-core run, per launch (rocprofv3 --pmc; tools/pmc_summary.py)"; python tools/pmc_summary.py $(ls $OUT/hm/*/*counter_collection.csv | head -1) | grep -A7 "hyb_conv_mfma";
  echo; echo "== all levels, HIP events (matrix cores / vector units)"; python tools/hybrid_ks_bench.py 2>/dev/null; DACAPO_HYB_MFMA=0 python tools/hybrid_ks_bench.py 2>/dev/null; } > $OUT/${R}_hybrid_ks_kernels.txt
rm -rf $OUT/hy $OUT/hv $OUT/hm
# B4b. VALU occupancy of the single-crossing NTT (what bounds it: profiles/r03_ntt_full.txt)
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/nv -- python3 $ROOT/tools/ntt_variant_only.py 1 4096 2 > /dev/null 2> $OUT/nv.err
cd $ROOT
python tools/ntt_valu.py $(ls $OUT/nv/*/*counter_collection.csv | head -1) $(ls $OUT/nv/*/*kernel_trace.csv | head -1) > $OUT/${R}_ntt_valu.json
rm -rf $OUT/nv
# B5. the single-crossing NTT against the two-launch tiles (HIP events)
for n in 512 1024 2048 4096; do python tools/ntt_full_check.py $n 20; done > $OUT/${R}_ntt_full_check.txt 2>/dev/null
# C. latency of dependent chains, D. several ciphertext streams on one GPU, E. the bench line itself, F. per-op table for the reference's planner
python tools/chain_bench.py > $OUT/${R}_chain_latency.txt 2>/dev/null
for s in 2 4 8; do python bench.py --streams $s --no-cpu-baseline --no-lowerings 2>/dev/null | python tools/bench_brief.py; done > $OUT/${R}_streams.txt
python tools/profile_backend.py --out $OUT/${R}_profiled_SEAL_MI355X.json > $OUT/profile_backend.log 2>&1
python bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err
tail -c 400 $OUT/${R}_bench.json
