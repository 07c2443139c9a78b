set -u
O=$GRAFT_REPO_ROOT/gpurun_out/x12; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in apart adjacent apart adjacent; do
  echo "== $v"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$v -- python3 $GRAFT_REPO_ROOT/tools/experiments/rows_twiddle_sharing.py $v 10 > /dev/null 2> $O/err_$v.txt
  python3 - <<PY
import csv,glob,re
f=glob.glob("$O/k_$v/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:3]:
    n=re.sub(r"\(.*","",r['Name']).replace('void dacapo::','')
    print(f"  {n:52s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f}")
PY
  rm -rf $O/k_$v
done
