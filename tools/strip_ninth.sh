#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
AB=fastdiffsr_amd/csrc/ab
timeout 1500 python -m pytest tests/test_gpu_strip.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/kernel_avg.sh bf16 64 'conv_strip' new= prev=$AB/libfdsr_hip_prev.so nofence3=$AB/libfdsr_hip_nofence3.so 2>&1 | grep -v "total kernel" | tee $O/kavg_stage3.txt
bash tools/kernel_avg.sh f16x3 16 'conv_strip' new= prev=$AB/libfdsr_hip_prev.so 2>&1 | grep -v "total kernel" | tee -a $O/kavg_stage3.txt
