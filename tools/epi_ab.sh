#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/epi; mkdir -p $O
cd $R
V=fastdiffsr_amd/csrc/ab/libfdsr_hip_oldepi.so
bash tools/kernel_avg.sh bf16 64 'conv_k32_kernel|conv_mfma_h_kernel<3, 1|conv_up2' new= old=$V 2>&1 | tee $O/kavg_bf16.txt
bash tools/lib_ab.sh $R/$V "" "" --precision bf16 --batch 64 --graph 2>&1 | tee $O/ab_bf16.txt
bash tools/lib_ab.sh $R/$V "strip=1" "strip=1" 2>&1 | tee $O/ab_f16x3.txt
timeout 900 python -m pytest tests/test_gpu_k32.py -x -q -k "layerwise_forced or small_workgroup" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
