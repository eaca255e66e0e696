# same-box perf-only A/B of variant libraries (tools/build_variant.sh) through tools/lib_probe.py:  bash tools/ab_probe.sh base tagA ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
for rep in 1 2; do
  for tag in "$@"; do
    FDSR_LIB=$R/fastdiffsr_amd/csrc/ab/libfdsr_hip_$tag.so timeout 300 python tools/lib_probe.py ${PREC:-f16x3} ${BATCH:-16} < /dev/null 2>$O/probe_${tag}_$rep.err | tee -a $O/probe_summary.txt
  done
done
