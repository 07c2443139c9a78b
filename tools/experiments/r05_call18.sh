set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05ae
for w in -1 0; do
  rm -rf /tmp/kt_$w
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$w -- python3 $R/tools/legs/per_op_only.py 20 --only rotate_hop --opt wide_tile_wgs=$w > /dev/null 2>/tmp/kt.err
  echo "== rotate_hop, wide_tile_wgs=$w"; python3 $R/tools/summarize/summarize_trace.py $(ls /tmp/kt_$w/*/*kernel_trace.csv | head -1) | grep -v rocclr | head -9
  rm -rf /tmp/kb_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb_$w -- python3 $R/tools/legs/headline_only.py 3 b13 --opt wide_tile_wgs=$w > /dev/null 2>/tmp/kb.err
  echo "== b13, wide_tile_wgs=$w"; head -6 $(ls /tmp/kb_$w/*/*kernel_stats.csv | head -1) | cut -c1-60,200-330
done > $R/gpurun_out/r05ae/wide_kernels.txt 2>&1
cat $R/gpurun_out/r05ae/wide_kernels.txt
