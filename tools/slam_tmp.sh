cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
one() { python bench.py --steps 10 --warmup 3 --batch $1 --graph --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%.2f img/s  %.2f ms' % (r['value'], r['ms_per_step']))"; }
echo -n "B=1: "; one 1
echo -n "B=1: "; one 1
bash tools/kernel_avg.sh f16x3 1 "slam|clam|conv_in8|conv_out3|posterior" tree= 2>&1 | grep -v "^$" | tail -8
