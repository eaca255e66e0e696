#!/bin/bash
# The column-strip conv's GPU loop in one call: parity tests, per-launch time of the strip kernels and of the tile kernels they replace
# (or stand beside) at bf16 B=64 and f16x3 B=16, under option sets / variant libraries:
#   gpurun -- 'bash tools/strip_check.sh [<label>=<lib.so or empty>[:opt=val,...] ...]'      (default: the tree, strip on / off)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_strip.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
[ $# -eq 0 ] && set -- on= off=:strip=0
bash tools/kernel_avg.sh bf16 64 'conv_strip|conv_k32_kernel<8, [24], 2|conv_mfma_h_kernel<3, 1, false, 16' "$@" 2>&1 | tee $O/kavg_bf16.txt
bash tools/kernel_avg.sh f16x3 16 'conv_strip|conv_k32_kernel<6, 2, 1' "$@" 2>&1 | tee $O/kavg_f16x3.txt
