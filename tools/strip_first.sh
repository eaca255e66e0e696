#!/bin/bash
# first GPU pass of the column-strip conv: parity tests, same-box A/B at bf16 B=64 (graph), per-kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/strip; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_strip.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
bash tools/ab_options.sh "strip=0" "strip=1" --precision bf16 --batch 64 --graph 2>&1 | tee $O/ab_bf16_b64.txt
bash tools/opt_stats.sh "strip=0" "strip=1" --precision bf16 --batch 64 --graph > $O/optstats.txt 2>&1
cp gpurun_out/optstats/cmp.txt $O/cmp_bf16_b64.txt; head -30 $O/cmp_bf16_b64.txt
