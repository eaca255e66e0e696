#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
bash tools/opt_stats.sh "strip=0" "strip=1" --precision bf16 --batch 64 --graph > $O/optstats.txt 2>&1
(head -12 gpurun_out/optstats/cmp.txt; tail -1 gpurun_out/optstats/cmp.txt) | tee $O/cmp_strip.txt
rocm-smi --showclocks 2>/dev/null | head -20
