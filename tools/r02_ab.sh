# A/B on one box: variant libraries (tools/build_variant.sh) through FDSR_LIB, interleaved twice
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
if [ "$RUN_TESTS" = "1" ]; then timeout 1500 python -m pytest tests -m gpu -q -s > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt; fi
for rep in 1 2; do
  for tag in $@; do
    FDSR_LIB=$R/fastdiffsr_amd/csrc/ab/libfdsr_hip_$tag.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sub-records > $O/ab_${tag}_$rep.json 2>$O/ab_${tag}_$rep.err
    python -c "import json;d=json.load(open('$O/ab_${tag}_$rep.json'));print('$tag',$rep,round(d['value'],2),d['roofline']['frac'])" | tee -a $O/ab_summary.txt
  done
done
for rep in 1 2; do
  FDSR_GN_TAIL=1 timeout 300 python bench.py --batch 1 --graph --steps 30 --warmup 5 --no-cpu-baseline --no-sub-records > $O/b1_tail_$rep.json 2>/dev/null
  FDSR_GN_TAIL=0 timeout 300 python bench.py --batch 1 --graph --steps 30 --warmup 5 --no-cpu-baseline --no-sub-records > $O/b1_notail_$rep.json 2>/dev/null
  python -c "import json;print('b1 tail',json.load(open('$O/b1_tail_$rep.json'))['value'],'notail',json.load(open('$O/b1_notail_$rep.json'))['value'])" | tee -a $O/ab_summary.txt
  FDSR_GN_TAIL=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sub-records > $O/b16_tail_$rep.json 2>/dev/null
  FDSR_GN_TAIL=0 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sub-records > $O/b16_notail_$rep.json 2>/dev/null
  python -c "import json;print('b16 tail',json.load(open('$O/b16_tail_$rep.json'))['value'],'notail',json.load(open('$O/b16_notail_$rep.json'))['value'])" | tee -a $O/ab_summary.txt
done
