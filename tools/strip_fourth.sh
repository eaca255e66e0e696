#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
AB=fastdiffsr_amd/csrc/ab
timeout 1500 python -m pytest tests/test_gpu_strip.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
bash tools/kernel_avg.sh bf16 64 'conv_strip' tree= none=$AB/libfdsr_hip_none.so noact=$AB/libfdsr_hip_noact.so 2>&1 | grep -v "total kernel" | tee $O/kavg_diag.txt
bash tools/k32_pmc.sh --precision bf16 --batch 64 > $O/pmc.txt 2>&1
grep -E "^kernel|conv_strip" $O/pmc.txt | cut -c1-220
