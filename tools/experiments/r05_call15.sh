#!/bin/bash
# round 5, call 15: kernel-time table of config 4 with lazy sums
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05q; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $ROOT/tools/legs/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 9 8 --opt hyb_lazy_sum=1 > $OUT/c4_lazy_prof.txt 2> $OUT/c4.err
cp $(ls $OUT/c4/*/*kernel_stats.csv | head -1) $OUT/config4_lazy_kernel_stats.csv
rm -rf $OUT/c4
head -30 $OUT/config4_lazy_kernel_stats.csv | cut -c1-200
