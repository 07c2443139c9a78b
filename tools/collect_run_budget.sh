#!/bin/bash
# Kernel-by-kernel budget of one run() of a lowering of the ResNet-20 trace (durations, HBM bytes, VALU instructions, floors):
#   gpurun --timeout 1500 -- 'bash tools/collect_run_budget.sh r05 b13 [--opt name=value ...]'      (lowering: headline | headline_s<streams> | b6 | b13)
# -> gpurun_out/<round>/<round>_run_budget_<lowering>.txt / .json (+ raw CSVs, gzipped)
set -u
R=${1:-r05}; LOW=${2:-b13}; shift; shift || true
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
kt() { ls $1/*/*kernel_trace.csv | head -1; }
cc() { ls $1/*/*counter_collection.csv | head -1; }
ARG=$LOW; [ $LOW = headline ] && ARG=""
case $LOW in headline_s*) ARG=""; EXTRA="--streams ${LOW#headline_s} $EXTRA";; esac   # headline_s16: 16 images per run() in one VM
CMD="python3 $ROOT/tools/legs/headline_only.py 3 $ARG $EXTRA"
D=$OUT/raw_run_$LOW; rm -rf $D; mkdir -p $D
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D/kt -- $CMD > /dev/null 2> $D/kt.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pf -- $CMD > /dev/null 2> $D/pf.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pw -- $CMD > /dev/null 2> $D/pw.err
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $D/pv -- $CMD > /dev/null 2> $D/pv.err
cp $(kt $D/kt) $D/kernel_trace.csv; cp $(cc $D/pf) $D/fetch.csv; cp $(cc $D/pw) $D/write.csv; cp $(cc $D/pv) $D/valu.csv
rm -rf $D/kt $D/pf $D/pw $D/pv
{ echo "One run() of the ResNet-20 trace, lowering $LOW, kernel by kernel: tools/collect_run_budget.sh $R $LOW $EXTRA"
  echo "library sha256: $(sha256sum $ROOT/dacapo_amd/lib/libSEAL_HEVM.so | cut -c1-64)"
  python3 $ROOT/tools/summarize/run_budget.py $D/kernel_trace.csv $D/fetch.csv $D/write.csv $D/valu.csv top=16 label=resnet20.$LOW json=$OUT/${R}_run_budget_$LOW.json; } > $OUT/${R}_run_budget_$LOW.txt 2> $D/budget.err
gzip -f $D/*.csv
cat $OUT/${R}_run_budget_$LOW.txt
