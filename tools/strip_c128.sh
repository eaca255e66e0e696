#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_strip.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
bash tools/kernel_avg.sh bf16 64 'conv_strip|conv_k32_kernel<8, 4, 2' s27=:strip=27 s91=:strip=91 2>&1 | tee $O/kavg_c128.txt
bash tools/ab_options.sh "strip=27" "strip=91" --precision bf16 --batch 64 --graph 2>&1 | tee $O/ab_c128.txt
