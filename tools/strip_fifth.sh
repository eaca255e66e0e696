#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
bash tools/ab_options.sh "strip=0" "strip=1" --precision bf16 --batch 64 --graph 2>&1 | tee $O/ab_bf16_b64.txt
bash tools/ab_options.sh "strip=1 rider=2" "strip=1 rider=3" --precision bf16 --batch 64 --graph 2>&1 | tee -a $O/ab_bf16_b64.txt
bash tools/opt_stats.sh "strip=1 rider=2" "strip=1 rider=3" --precision bf16 --batch 64 --graph > $O/optstats.txt 2>&1
head -24 gpurun_out/optstats/cmp.txt | tee $O/cmp_rider.txt
