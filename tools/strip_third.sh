#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_strip.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
bash tools/kernel_avg.sh bf16 64 'conv_strip|conv_k32_kernel<8, 2' tree= "$@" 2>&1 | tee $O/kavg.txt
bash tools/k32_pmc.sh --precision bf16 --batch 64 > $O/pmc.txt 2>&1
grep -E "^kernel|conv_strip|conv_k32_kernel<8, 2" $O/pmc.txt | cut -c1-220
