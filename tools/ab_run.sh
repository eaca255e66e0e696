# same-box A/B of variant libraries built by tools/build_variant.sh:  bash tools/ab_run.sh base tagA tagB ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
for rep in 1 2; do
  for tag in "$@"; do
    FDSR_LIB=$R/fastdiffsr_amd/csrc/ab/libfdsr_hip_$tag.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sub-records > $O/ab_${tag}_$rep.json 2>$O/ab_${tag}_$rep.err < /dev/null
    python -c "import json;d=json.load(open('$O/ab_${tag}_$rep.json'));print('$tag',$rep,round(d['value'],2),round(d['roofline']['frac'],4))" | tee -a $O/ab_summary.txt
  done
done
