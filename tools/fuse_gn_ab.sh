#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fusegn; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fuse_gn.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
bash tools/ab_options.sh "fuse_gn=0" "fuse_gn=1" --batch 1 --graph --steps 20 --warmup 5 2>&1 | tee $O/ab_b1.txt
bash tools/ab_options.sh "fuse_gn=0" "fuse_gn=1" --batch 4 --graph 2>&1 | tee $O/ab_b4.txt
