# same-box A/B of the f16x3 weight-gradient forms (training step, B=32):  gpurun -- 'bash tools/wg_ab.sh'
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x < /dev/null 2>&1 | tail -3
run() { timeout 300 python bench.py --train --precision f16x3 --steps 3 --warmup 1 $2 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1))"; }
for r in 1 2; do
run "4-wave                      " "--debug-option wgrad_form=1"
run "8-wave                      " "--debug-option wgrad_form=2"
run "8-wave in-row               " "--debug-option wgrad_colsum=0"
run "8-wave in-row + column sums " ""
done
