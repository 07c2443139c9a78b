#!/bin/bash
# One grouped-digit rotation hop at N = 2^17, level 31: HIP-event times at all levels + measured HBM bytes per kernel (FETCH_SIZE / WRITE_SIZE
# passes), for the default launch sequence.  usage (on the GPU box): bash tools/experiments/hop_bytes.sh <outdir> [--opt name=value ...]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/${1:-hop}; shift
mkdir -p $OUT
export TMPDIR=/tmp
kt() { ls $1/*/*kernel_trace.csv | head -1; }
cc() { ls $1/*/*counter_collection.csv | head -1; }
cd /tmp
python3 $ROOT/tools/hybrid_ks_bench.py 17 39 8 7 10 0 "$@" > $OUT/levels.json 2> $OUT/err.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/hy -- python3 $ROOT/tools/hybrid_ks_bench.py 17 39 8 7 10 31 "$@" > $OUT/hop.json 2>> $OUT/err.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/hyf -- python3 $ROOT/tools/hybrid_ks_bench.py 17 39 8 7 10 31 "$@" > /dev/null 2>> $OUT/err.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/hyw -- python3 $ROOT/tools/hybrid_ks_bench.py 17 39 8 7 10 31 "$@" > /dev/null 2>> $OUT/err.txt
python3 $ROOT/tools/kernel_bytes.py $(kt $OUT/hy) $(cc $OUT/hyf) $(cc $OUT/hyw) per="hyb_mac_kernel<0>" top=16 > $OUT/bytes.txt
rm -rf $OUT/hy $OUT/hyf $OUT/hyw
cat $OUT/levels.json | python3 -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"N\"'):
        d=json.loads(ln); print('hop us by level:', ' / '.join('%d: %.0f' % (l['level'], l['hop_us']) for l in d['levels']))"
cat $OUT/bytes.txt
