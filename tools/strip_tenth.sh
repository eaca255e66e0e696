#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
AB=fastdiffsr_amd/csrc/ab
bash tools/kernel_avg.sh bf16 64 'conv_strip' xs4= xs6=$AB/libfdsr_hip_xs6.so xs8=$AB/libfdsr_hip_xs8.so xs12=$AB/libfdsr_hip_xs12.so 2>&1 | grep -v "total kernel\|4, true, 2\|4, false, 2" | tee $O/kavg_xs1.txt
bash tools/kernel_avg.sh f16x3 16 'conv_strip' xs4= xs6=$AB/libfdsr_hip_xs6.so xs8=$AB/libfdsr_hip_xs8.so 2>&1 | grep -v "total kernel" | tee -a $O/kavg_xs1.txt
