# Collect the round's evidence on the GPU box into gpurun_out/refresh (copy what is to be judged into profiles/):
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh'
# Every step is bounded by `timeout` and reads nothing from stdin; counters are collected in runs of their own.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 < /dev/null
# the driver's command, as the driver runs it
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err < /dev/null
timeout 600 python bench.py --train --precision f16x3 --steps 3 --warmup 1 > $O/bench_train_f16x3_b32.json 2>/dev/null < /dev/null
timeout 600 python bench.py --train --precision f32 --steps 3 --warmup 1 > $O/bench_train_f32_b32.json 2>/dev/null < /dev/null
cd /tmp && export TMPDIR=/tmp
H="--no-cpu-baseline --no-sub-records"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b16 -- python3 $R/bench.py --steps 20 --warmup 5 $H > $O/stats.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -o t -- python3 $R/bench.py --train --precision f16x3 --steps 2 --warmup 1 > $O/stats_train.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train32 -o t -- python3 $R/bench.py --train --precision f32 --steps 2 --warmup 1 > $O/stats_train32.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/pmc_fetch.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/pmc_write.log 2>&1 < /dev/null
# MFMA / VALU / LDS utilisation: two more counter passes (<= 8 SQ counters each), summarised by tools/pmc_util.py
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/util_a -o a -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/util_a.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/util_b -o b -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/util_b.log 2>&1 < /dev/null
# the training step (f16x3, B=32): the same two utilisation passes
T="--train --precision f16x3"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/tutil_a -o a -- python3 $R/bench.py $T --steps 1 --warmup 0 > $O/tutil_a.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/tutil_b -o b -- python3 $R/bench.py $T --steps 1 --warmup 0 > $O/tutil_b.log 2>&1 < /dev/null
cd $R
python tools/pmc_util.py $O/tutil_a $O/tutil_b 24 > $O/pmc_mfma_util_train_f16x3_b32.txt 2> $O/pmc_tutil.err < /dev/null
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_hbm_traffic_f16x3_b16.json 2> $O/pmc_traffic.err < /dev/null
python tools/pmc_util.py $O/util_a $O/util_b 16 > $O/pmc_mfma_util_f16x3_b16.txt 2> $O/pmc_util.err < /dev/null
# keep the merge small: the raw counter dumps stay on the box
rm -rf $O/pmc_fetch/*/ $O/pmc_write/*/ 2>/dev/null
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
ls -la $O
echo refresh-done
