# same-box A/B of the res_conv rider (FDSR_RIDER=0 off, 1 default rule, 2 every res_conv):  gpurun -- 'bash tools/rider_ab.sh'
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_c_abi.py -m gpu -q -x < /dev/null 2>&1 | tail -3
for r in 1 2; do for m in 0 1 2; do
FDSR_RIDER=$m REPS=2 timeout 300 python tools/lib_probe.py f16x3 16 2>/dev/null < /dev/null | tail -1 | sed "s/^/rider=$m /"
done; done
for m in 0 1 2; do
FDSR_RIDER=$m REPS=2 timeout 300 python tools/lib_probe.py bf16 64 2>/dev/null < /dev/null | tail -1 | sed "s/^/rider=$m /"
FDSR_RIDER=$m REPS=2 timeout 300 python tools/lib_probe.py f16x3 1 2>/dev/null < /dev/null | tail -1 | sed "s/^/rider=$m /"
done
