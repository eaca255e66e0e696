cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_driver.py -m gpu -q -x < /dev/null 2>&1 | tail -3
for r in 1 2; do
timeout 300 python bench.py --train --precision f16x3 --steps 3 --warmup 1 --debug-option rider=0 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f16x3 rider off', round(d['value'],1))"
timeout 300 python bench.py --train --precision f16x3 --steps 3 --warmup 1 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f16x3 rider on ', round(d['value'],1))"
done
