set -u
O=$GRAFT_REPO_ROOT/gpurun_out/x7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hevm.py -x -q -m gpu -k "sums or n_ary or conv" 2>&1 | tail -15
for g in 2048 -1; do
  echo "== sum_group_min_wgs=$g"
  timeout 300 python tools/boot_demo.py 17 5 1 14 8 7 --opt sum_group_min_wgs=$g --opt trace=1 2>&1 | grep -E "sums sharing|bootstrap:|decrypted"
  timeout 300 python tools/experiments/quick_headline.py 10 resnet20 --opt sum_group_min_wgs=$g 2>&1 | tail -1
done
timeout 600 python tools/resnet_real_boot.py 1 resnet20_nt16 17 1 b14 8 7 --opt trace=1 > $O/c4.txt 2>&1; grep -E "sums sharing|sums, in limbs" $O/c4.txt; tail -1 $O/c4.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['run_s'], d['first_run_s'], d['rms_vs_torch'])"
