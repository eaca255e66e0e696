set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q > $O/train_tests.txt 2>&1 < /dev/null; tail -12 $O/train_tests.txt
for prec in f16x3 f32; do
timeout 600 python bench.py --train --precision $prec --steps 3 --warmup 1 > $O/train_b32_$prec.json 2> $O/train_b32_$prec.err < /dev/null; cut -c1-260 $O/train_b32_$prec.json
done
if [ "$PROF" = "1" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- python3 $R/bench.py --train --steps 2 --warmup 1 > $O/stats.log 2>&1 < /dev/null
fi
echo done
