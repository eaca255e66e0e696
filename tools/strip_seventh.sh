#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_strip.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
bash tools/kernel_avg.sh f16x3 16 'conv_strip|conv_k32_kernel<6, 2' s1=:strip=1 s3=:strip=3 2>&1 | tee $O/kavg_f16.txt
bash tools/ab_options.sh "strip=1" "strip=3" 2>&1 | tee $O/ab_f16x3_b16.txt
