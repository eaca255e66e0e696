#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
bash tools/kernel_avg.sh bf16 64 'conv_strip' lb2=:strip=1 lb1=:strip=5 2>&1 | grep -v "total kernel" | tee $O/kavg_lb1.txt
