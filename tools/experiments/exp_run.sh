set -u
O=$GRAFT_REPO_ROOT/gpurun_out/x9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hevm.py -x -q -m gpu -k "sums or n_ary" 2>&1 | tail -2
for v in default rr default rr; do
  if [ $v = default ]; then unset DACAPO_AMD_LIB; else export DACAPO_AMD_LIB=$GRAFT_REPO_ROOT/dacapo_amd/lib/variants/libSEAL_HEVM.$v.so; fi
  echo "== $v"
  timeout 300 python3 tools/hybrid_ks_bench.py 17 39 8 7 10 0 2>/dev/null | python3 -c "import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"N\"'):
        d=json.loads(ln); print('hop us by level:', ' / '.join('%d: %.0f' % (l['level'], l['hop_us']) for l in d['levels']))"
  timeout 300 python tools/boot_demo.py 17 5 1 14 8 7 2>&1 | grep -E "bootstrap:"
done
