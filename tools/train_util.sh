# kernel stats + MFMA/VALU/LDS utilisation of the training step (f16x3, B=32):  gpurun -- 'bash tools/train_util.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/train_util; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="--train --precision f16x3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- python3 $R/bench.py $T --steps 2 --warmup 1 > $O/stats.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/util_a -o a -- python3 $R/bench.py $T --steps 1 --warmup 0 > $O/util_a.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/util_b -o b -- python3 $R/bench.py $T --steps 1 --warmup 0 > $O/util_b.log 2>&1 < /dev/null
cd $R
python tools/pmc_util.py $O/util_a $O/util_b 24 > $O/pmc_util_train_f16x3_b32.txt 2> $O/pmc_util.err < /dev/null
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
cat $O/pmc_util_train_f16x3_b32.txt
