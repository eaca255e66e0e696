#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_strip.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
bash tools/kernel_avg.sh bf16 64 'conv_strip|conv_k32_kernel<8, 2|conv_mfma_h_kernel<3, 1, false, 16' s3=:strip=3 s27=:strip=27 s59=:strip=59 2>&1 | tee $O/kavg_wide.txt
bash tools/ab_options.sh "strip=3" "strip=27" --precision bf16 --batch 64 --graph 2>&1 | tee $O/ab_wide.txt
