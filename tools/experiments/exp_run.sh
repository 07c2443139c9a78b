set -u
O=$GRAFT_REPO_ROOT/gpurun_out/x8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config4_geometry.py tests/test_gpu_hybrid.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/boot_demo.py 17 5 1 14 8 7 2>&1 | grep -E "bootstrap:|decrypted"
timeout 300 python3 tools/hybrid_ks_bench.py 17 39 8 7 10 0 2>/dev/null | python3 -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"N\"'):
        d=json.loads(ln); print('hop us by level:', ' / '.join('%d: %.0f' % (l['level'], l['hop_us']) for l in d['levels']))"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -- python3 $GRAFT_REPO_ROOT/tools/boot_demo.py 17 5 1 14 8 7 > /dev/null 2> $O/b.err
python3 - <<PY
import csv,glob,re
f=glob.glob("$O/b/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n=re.sub(r"\(.*","",r['Name']).replace('void dacapo::','')
    print(f"  {n:52s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} {int(r['TotalDurationNs'])/1e6:8.1f}")
PY
rm -rf $O/b
cd $GRAFT_REPO_ROOT
timeout 600 python tools/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 8 7 > $O/c4.txt 2>&1; tail -1 $O/c4.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['run_s'], d['first_run_s'], d['rms_vs_torch'])"
