cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x < /dev/null 2>&1 | tail -5
for r in 1 2; do
FDSR_WGRAD_H4=1 timeout 300 python bench.py --train --precision f16x3 --steps 3 --warmup 1 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('h4', d['value'])"
timeout 300 python bench.py --train --precision f16x3 --steps 3 --warmup 1 2>/dev/null < /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('h8', d['value'])"
done
