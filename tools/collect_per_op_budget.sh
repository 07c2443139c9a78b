#!/bin/bash
# Per-kernel budget (durations, HBM bytes, VALU instructions, floors) of the single ops at 13 primes and of config 3, one op per process:
#   gpurun --timeout 1500 -- 'bash tools/collect_per_op_budget.sh r05 [--opt name=value ...]'
# -> gpurun_out/<round>/<round>_per_op_kernel_bytes.txt (+ .json per op, + the raw CSVs under raw_<op>/ for re-processing)
set -u
R=${1:-r05}; shift || true
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IT=20
kt() { ls $1/*/*kernel_trace.csv | head -1; }
cc() { ls $1/*/*counter_collection.csv | head -1; }
TXT=$OUT/${R}_per_op_kernel_bytes.txt
{ echo "Single ops at 13 primes (N = 2^15) and config 3 (N = 2^16, 24 + 1 primes), kernel by kernel: tools/collect_per_op_budget.sh $R $EXTRA"
  echo "library sha256: $(sha256sum $ROOT/dacapo_amd/lib/libSEAL_HEVM.so | cut -c1-64)"
  echo "columns: tools/summarize/per_op_budget.py (floors: bytes / 5.5 TB/s; SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.05 GHz, then x the grid's quantisation)"; echo; } > $TXT
for op in rotate_hop mulcc_relin rescale cfg3; do
  D=$OUT/raw_$op; rm -rf $D; mkdir -p $D
  CMD="python3 $ROOT/tools/legs/per_op_only.py $IT --only $op $EXTRA"
  timeout 300 $CMD > $D/events.json 2> $D/events.err
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $D/kt -- $CMD > /dev/null 2> $D/kt.err
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pf -- $CMD > /dev/null 2> $D/pf.err
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pw -- $CMD > /dev/null 2> $D/pw.err
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/pv -- $CMD > /dev/null 2> $D/pv.err
  EV=$(python3 -c "
import json,sys
d=json.loads(open('$D/events.json').read().strip().splitlines()[-1])
print(d['cfg3']['us'] if '$op'=='cfg3' else d['per_op_13_primes']['$op']['us'])" 2>/dev/null || echo 0)
  IT2=$IT; EX=""; [ $op = cfg3 ] && IT2=10
  cp $(kt $D/kt) $D/kernel_trace.csv; cp $(cc $D/pf) $D/fetch.csv; cp $(cc $D/pw) $D/write.csv; cp $(cc $D/pv) $D/valu.csv
  rm -rf $D/kt $D/pf $D/pw $D/pv
  python3 $ROOT/tools/summarize/per_op_budget.py $op $IT2 $D/kernel_trace.csv $D/fetch.csv $D/write.csv $D/valu.csv event_us=$EV json=$OUT/${R}_per_op_budget_$op.json $EX >> $TXT 2>> $D/budget.err
  echo >> $TXT
  gzip -f $D/*.csv
done
tail -60 $TXT
