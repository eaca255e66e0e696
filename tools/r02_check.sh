# round-2 GPU check: parity tests (optionally a subset: TESTS="tests/test_gpu_train.py"), then the driver's bench command
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 2400 python -m pytest ${TESTS:-tests} -m gpu -q -s > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
if [ "$BENCH" != "0" ]; then
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 600 $O/bench_driver_cmd.err
fi
