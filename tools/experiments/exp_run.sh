set -u
O=$GRAFT_REPO_ROOT/gpurun_out/x5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 $GRAFT_REPO_ROOT/tools/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 8 7 > $O/c4.txt 2> $O/c4.err
cp $(ls $O/c4/*/*kernel_stats.csv | head -1) $O/c4_kernel_stats.csv; rm -rf $O/c4
tail -1 $O/c4.txt | cut -c1-400
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
