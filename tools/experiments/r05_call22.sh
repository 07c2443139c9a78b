#!/bin/bash
# round 5, call 22: per-kernel bytes of one bootstrap with ROWS phases prime by prime (the B3 block of tools/collect_profiles.sh)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05q; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
kt() { ls $1/*/*kernel_trace.csv | head -1; }
cc() { ls $1/*/*counter_collection.csv | head -1; }
BT="$ROOT/tools/legs/boot_demo.py 17 5 1 14 9 8 --opt hyb_lazy_sum=1"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/bt -- python3 $BT > $OUT/bt.txt 2> $OUT/bt.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/btf -- python3 $BT --opt plan_graph=0 > /dev/null 2> $OUT/btf.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/btw -- python3 $BT --opt plan_graph=0 > /dev/null 2> $OUT/btw.err
cd $ROOT
{ grep -E "bootstrap:|decrypted" $OUT/bt.txt; python tools/summarize/kernel_bytes.py $(kt $OUT/bt) $(cc $OUT/btf) $(cc $OUT/btw) top=24; } > $OUT/boot_bytes_prime_major.txt
rm -rf $OUT/bt $OUT/btf $OUT/btw
cat $OUT/boot_bytes_prime_major.txt | cut -c1-200
