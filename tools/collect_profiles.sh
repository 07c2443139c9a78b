#!/bin/bash
# Everything under profiles/ for one round, collected on the GPU box in one gpurun call:
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh r06'
# Outputs go to gpurun_out/<round>/ (merged back by gpurun); copy the summaries into profiles/ afterwards.
# rocprofv3 runs with the program directly after `--`, counters (--pmc) in their own passes, kernel-trace / stats only.
set -u
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
kt() { ls $1/*/*kernel_trace.csv | head -1; }
cc() { ls $1/*/*counter_collection.csv | head -1; }
# A. kernel trace + stats of the bench command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lowerings --no-config4 > $OUT/bench_under_profiler.json 2> $OUT/kt.err
cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/${R}_kernel_stats.csv
cd $ROOT
python tools/summarize/summarize_trace.py $(kt $OUT/kt) > $OUT/${R}_by_kernel_and_grid.txt
python tools/summarize/summarize_trace.py $(kt $OUT/kt) 4096 > $OUT/${R}_roofline_leg_launches.txt
python tools/summarize/timeline_gaps.py $(kt $OUT/kt) > $OUT/${R}_timeline.txt
python tools/summarize/top_kernels.py $(kt $OUT/kt) > $OUT/${R}_top_kernels.json
rm -rf $OUT/kt
# B. HBM traffic of the NTT leg: FETCH_SIZE and WRITE_SIZE need separate passes
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pf -- python3 $ROOT/tools/legs/ntt_only.py 15 4096 2 > /dev/null 2> $OUT/pf.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pw -- python3 $ROOT/tools/legs/ntt_only.py 15 4096 2 > /dev/null 2> $OUT/pw.err
cd $ROOT
python tools/summarize/collect_traffic.py $(cc $OUT/pf) $(cc $OUT/pw) > $OUT/${R}_ntt_hbm_traffic.json
rm -rf $OUT/pf $OUT/pw
# B2. the timed step's own kernels: durations, FETCH_SIZE / WRITE_SIZE per kernel, algorithmic bytes where the grid encodes (level, batch)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/hk -- python3 $ROOT/tools/legs/headline_only.py 3 > /dev/null 2> $OUT/hk.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/hf -- python3 $ROOT/tools/legs/headline_only.py 3 > /dev/null 2> $OUT/hf.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/hw -- python3 $ROOT/tools/legs/headline_only.py 3 > /dev/null 2> $OUT/hw.err
cd $ROOT
python tools/summarize/kernel_traffic.py $(kt $OUT/hk) $(cc $OUT/hf) $(cc $OUT/hw) > $OUT/${R}_step_kernels.json
rm -rf $OUT/hk $OUT/hf $OUT/hw
# B3. config 4 (N = 2^17, 38 real bootstraps, grouped-digit keys): kernel-time table of the whole run; measured HBM bytes per kernel on ONE
#     bootstrap of the same geometry (rocprofv3 --pmc on the whole config-4 program crashes or hangs: profiles/r04_experiments.txt item 12)
cd /tmp
C4="$ROOT/tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8"   # as bench.py's config-4 leg runs it: the library's default options (round 6)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $C4 > $OUT/${R}_config4_under_profiler.txt 2> $OUT/c4.err
cp $(ls $OUT/c4/*/*kernel_stats.csv | head -1) $OUT/${R}_config4_kernel_stats.csv
rm -rf $OUT/c4
BT="$ROOT/tools/legs/boot_demo.py 17 5 1 14 9 8"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/bt -- python3 $BT > $OUT/bt.txt 2> $OUT/bt.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/btf -- python3 $BT --opt plan_graph=0 > /dev/null 2> $OUT/btf.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/btw -- python3 $BT --opt plan_graph=0 > /dev/null 2> $OUT/btw.err
cd $ROOT
{ echo "one real bootstrap at config 4's geometry (N = 2^17, 31 + 9 primes, digits of 8, 1 -> 14 primes): python tools/legs/boot_demo.py 17 5 1 14 9 8 (default options, as config 4's headline runs)"
  echo "the process = key generation + encoding + 3 runs; bytes = FETCH_SIZE x 2 + WRITE_SIZE per kernel (plan run launch by launch for the counters)"
  grep -E "bootstrap:|decrypted" $OUT/bt.txt
  python tools/summarize/kernel_bytes.py $(kt $OUT/bt) $(cc $OUT/btf) $(cc $OUT/btw) top=30; } > $OUT/${R}_boot_kernel_bytes.txt
rm -rf $OUT/bt $OUT/btf $OUT/btw
# B4. one grouped-digit key switch at N = 2^17, top level: every kernel of the sequence with its measured HBM bytes per hop, for the fused
#     sequence (default, hyb_fuse = 2), the loader form (1) and round 3's sequence (0); the matrix-core counters of the default
cd /tmp
{ for f in 2 1 0; do
    timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/hy$f -- python3 $ROOT/tools/legs/hybrid_ks_bench.py 17 40 9 8 10 31 --opt hyb_fuse=$f > $OUT/hy${f}_hop.json 2> $OUT/hy.err
    timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/hyf$f -- python3 $ROOT/tools/legs/hybrid_ks_bench.py 17 40 9 8 10 31 --opt hyb_fuse=$f > /dev/null 2>> $OUT/hy.err
    timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/hyw$f -- python3 $ROOT/tools/legs/hybrid_ks_bench.py 17 40 9 8 10 31 --opt hyb_fuse=$f > /dev/null 2>> $OUT/hy.err
    echo "== hyb_fuse = $f: one rotation hop at N = 2^17, level 31 (4 digits of 8 primes, 9 special primes); per hop = per launch of hyb_mac_kernel<0>"
    python3 -c "import json;d=json.loads(open('$OUT/hy${f}_hop.json').read().strip().splitlines()[-1])['levels'][0];print('HIP events, unprofiled loop:', d['hop_us'], 'us per hop;', d['ntt_equivalents'], 'NTT-equivalents; algorithmic bytes (tools/legs/hybrid_ks_bench.py)', d['algorithmic_bytes'])"
    python3 $ROOT/tools/summarize/kernel_bytes.py $(kt $OUT/hy$f) $(cc $OUT/hyf$f) $(cc $OUT/hyw$f) per="hyb_mac_kernel<0>" top=16
    echo
    rm -rf $OUT/hy$f $OUT/hyf$f $OUT/hyw$f
  done
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/hm -- python3 $ROOT/tools/legs/hybrid_ks_bench.py 17 40 9 8 5 31 > /dev/null 2> $OUT/hm.err
  echo "== counters of the matrix-core conversions, per launch (rocprofv3 --pmc; tools/summarize/pmc_summary.py)"; python3 $ROOT/tools/summarize/pmc_summary.py $(cc $OUT/hm) | grep -A7 "hyb_conv_mfma"
  rm -rf $OUT/hm
  echo; echo "== all levels, HIP events, rounds 3-4's key shape (5 digits of 7 under 8 special primes), hyb_fuse = 2"
  python3 $ROOT/tools/legs/hybrid_ks_bench.py 17 39 8 7 10 0 2>/dev/null
  echo; echo "== all levels, HIP events: hyb_fuse = 2 / 1 / 0"
  for f in 2 1 0; do python3 $ROOT/tools/legs/hybrid_ks_bench.py 17 40 9 8 10 0 --opt hyb_fuse=$f 2>/dev/null; done
} > $OUT/${R}_hybrid_ks_kernels.txt
cd $ROOT
# B4b. VALU occupancy of the single-crossing NTT
cd /tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/nv -- python3 $ROOT/tools/legs/ntt_variant_only.py 1 4096 2 > /dev/null 2> $OUT/nv.err
cd $ROOT
python tools/summarize/ntt_valu.py $(cc $OUT/nv) $(kt $OUT/nv) > $OUT/${R}_ntt_valu.json
rm -rf $OUT/nv
# B5. single ops at the reference's top level + config 3, kernel by kernel; the dataflow graph's width; chain latency; streams
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/po -- python3 $ROOT/tools/legs/per_op_only.py 20 > $OUT/${R}_per_op.json 2> $OUT/po.err
cp $(ls $OUT/po/*/*kernel_stats.csv | head -1) $OUT/${R}_per_op_kernel_stats.csv
rm -rf $OUT/po
cd $ROOT
for n in 512 640 768 900 1024 1300 1536 2048 4096; do python tools/legs/ntt_full_check.py $n 20; done > $OUT/${R}_ntt_full_check.txt 2>/dev/null
# B6 (round 5). kernel-by-kernel budgets: time, HBM bytes, VALU instructions and floors of the single ops and of one run() of the 13-prime lowering / the headline
bash tools/collect_per_op_budget.sh $R > $OUT/per_op_budget.log 2>&1
bash tools/collect_run_budget.sh $R b13 > $OUT/run_budget_b13.log 2>&1
bash tools/collect_run_budget.sh $R headline > $OUT/run_budget_headline.log 2>&1
python tools/legs/lowering_sweep.py 6 ks_items_fast=0 cols_pairs=0 tiny_tile_wgs=512 ks_items_fast=0,cols_pairs=0,tiny_tile_wgs=512,ntt_full_inv_pairs=0 > $OUT/${R}_lowering_sweep.txt 2>&1
python tools/legs/per_op_sweep.py 30 ks_items_fast=0 cols_pairs=0 tiny_tile_wgs=512 cols_pairs=0,tiny_tile_wgs=512 > $OUT/${R}_per_op_sweep.txt 2>&1
python tools/legs/chain_bench.py > $OUT/${R}_chain_latency.txt 2>/dev/null
# (round 5: the streams table is a leg of the bench line itself: bench.py streams_leg)
python tools/legs/profile_backend.py --out $OUT/${R}_profiled_SEAL_MI355X.json > $OUT/profile_backend.log 2>&1
# E. the bench line itself: it reports the round's sha-gated records (traffic, VALU counters, step kernels) when they were collected on the
#    library it times -- the ones above were, so they go to profiles/ (of this copy of the repo) first
for f in ntt_hbm_traffic ntt_valu step_kernels per_op_budget_rotate_hop per_op_budget_cfg3; do [ -s $OUT/${R}_$f.json ] && cp $OUT/${R}_$f.json $ROOT/profiles/; done
# (round 6) stdout's last line = the compact line the driver parses (<= 4 KB); the full record -- every leg, with --full the A/B legs too -- is its own file
python bench.py --full --out $OUT/${R}_bench_full.json > $OUT/${R}_bench_line.json 2> $OUT/bench.err
# F. the default run exactly as the driver starts it (timed), and the perf guard's figures by the guard's own code
S0=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 --out $OUT/${R}_bench_default_full.json > $OUT/${R}_bench_default_line.json 2> $OUT/bench_default.err
echo "default bench.py run: $(( $(date +%s) - S0 )) s" > $OUT/${R}_bench_default_seconds.txt
DACAPO_AMD_HOOKS= python tests/test_gpu_perf_guard.py --record $OUT/perf_guard.json $R > $OUT/perf_guard.log 2>&1
tail -c 2500 $OUT/${R}_bench_line.json
