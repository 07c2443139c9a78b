set -u
O=$GRAFT_REPO_ROOT/gpurun_out/x10; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in default s64 default s64; do
  if [ $v = default ]; then unset DACAPO_AMD_LIB; else export DACAPO_AMD_LIB=$GRAFT_REPO_ROOT/dacapo_amd/lib/variants/libSEAL_HEVM.$v.so; fi
  echo "== $v"
  timeout 300 python3 $GRAFT_REPO_ROOT/tools/hybrid_ks_bench.py 17 39 8 7 10 0 2>/dev/null | python3 -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"N\"'):
        d=json.loads(ln); print('hop us by level:', ' / '.join('%d: %.0f' % (l['level'], l['hop_us']) for l in d['levels']))"
  timeout 300 python3 $GRAFT_REPO_ROOT/tools/boot_demo.py 17 5 1 14 8 7 2>&1 | grep -E "bootstrap:"
done
export DACAPO_AMD_LIB=$GRAFT_REPO_ROOT/dacapo_amd/lib/variants/libSEAL_HEVM.s64.so
timeout 600 python3 -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_hybrid.py $GRAFT_REPO_ROOT/tests/test_gpu_config4_geometry.py -x -q -m gpu 2>&1 | tail -2
unset DACAPO_AMD_LIB
for o in 1024 100000; do echo "== ks_merge_special_min_wgs=$o"; timeout 300 python3 $GRAFT_REPO_ROOT/tools/per_op_only.py 30 --opt ks_merge_special_min_wgs=$o 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['per_op_13_primes']; print('per op', p['rotate_hop']['us'], p['mulcc_relin']['us'], p['rescale']['us'], 'cfg3', d['cfg3']['us'], d['cfg3'].get('grouped_digit_keys',{}).get('us'))"; done
