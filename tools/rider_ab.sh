# same-box A/B of the res_conv rider (debug option rider = 0 off, 1 bandwidth-bound ones, 2 every res_conv):  gpurun -- 'bash tools/rider_ab.sh'
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_c_abi.py -m gpu -q -x < /dev/null 2>&1 | tail -3
for r in 1 2; do for m in 0 1 2; do
REPS=2 PROBE_OPTS=rider=$m timeout 300 python tools/lib_probe.py f16x3 16 2>/dev/null < /dev/null | tail -1 | sed "s/^/rider=$m /"
done; done
for m in 0 1 2; do
REPS=2 PROBE_OPTS=rider=$m timeout 300 python tools/lib_probe.py bf16 64 2>/dev/null < /dev/null | tail -1 | sed "s/^/rider=$m /"
REPS=2 PROBE_OPTS=rider=$m timeout 300 python tools/lib_probe.py f16x3 1 2>/dev/null < /dev/null | tail -1 | sed "s/^/rider=$m /"
done
