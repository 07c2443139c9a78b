set -u
O=$GRAFT_REPO_ROOT/gpurun_out/x14; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in default pfall default pfall; do
  if [ $v = default ]; then unset DACAPO_AMD_LIB; else export DACAPO_AMD_LIB=$GRAFT_REPO_ROOT/dacapo_amd/lib/variants/libSEAL_HEVM.$v.so; fi
  echo "== $v"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$v -- python3 $GRAFT_REPO_ROOT/tools/ntt_only.py 17 160 10 > /dev/null 2> $O/err_$v.txt
  python3 - <<PY
import csv,glob,re
f=glob.glob("$O/k_$v/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    n=re.sub(r"\(.*","",r['Name']).replace('void dacapo::','')
    print(f"  {n:52s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f}")
PY
  rm -rf $O/k_$v
  timeout 300 python3 $GRAFT_REPO_ROOT/tools/hybrid_ks_bench.py 17 39 8 7 10 0 2>/dev/null | python3 -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"N\"'):
        d=json.loads(ln); print('hop us by level:', ' / '.join('%d: %.0f' % (l['level'], l['hop_us']) for l in d['levels']))"
done
