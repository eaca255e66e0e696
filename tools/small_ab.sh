#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/small; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_small.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for o in "small=0" "small=1" "small=1 small_max_wgs=256" "small=1 small_max_wgs=4096"; do
  for B in 1 4; do
    args=""; for kv in $o; do args="$args --debug-option $kv"; done
    v=$(python bench.py --batch $B --graph --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records --no-profile $args 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%.2f img/s  %.2f ms/step' % (r['value'], r['ms_per_step']))")
    echo "[B=$B $o] $v"
  done
done 2>&1 | tee $O/ab.txt
bash tools/kernel_avg.sh f16x3 1 'conv_small|conv_k32_kernel<2, 8|conv_k32_kernel<4, 8|splitk_reduce' off=:small=0 on=:small=1 2>&1 | tee $O/kavg_b1.txt
