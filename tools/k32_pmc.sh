#!/bin/bash
# PMC utilisation / wave-cycle breakdown of one f16x3 B=16 forward loop with the 16x16x32 form (same passes as refresh_profiles_r03.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k32pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
H="--no-cpu-baseline --no-sub-records --no-profile"
PA="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM"
PB="GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
PC="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
n=f16x3_b16
timeout 600 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/util_a_$n -o a -- python3 $R/bench.py --steps 1 --warmup 0 $H "$@" > $O/util_a_$n.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/util_b_$n -o b -- python3 $R/bench.py --steps 1 --warmup 0 $H "$@" > $O/util_b_$n.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc $PC --kernel-trace --output-format csv -d $O/util_c_$n -o c -- python3 $R/bench.py --steps 1 --warmup 0 $H "$@" > $O/util_c_$n.log 2>&1 < /dev/null
cd $R
python tools/pmc_util.py $O/util_a_$n $O/util_b_$n 16 > $O/pmc_mfma_util_$n.txt 2> $O/pmc_util_$n.err < /dev/null
python tools/pmc_wave.py $O/util_c_$n $O/util_a_$n 12 > $O/pmc_wave_cycles_$n.txt 2> $O/pmc_wave_$n.err < /dev/null
find $O -name "*counter_collection.csv" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
cat $O/pmc_mfma_util_$n.txt $O/pmc_wave_cycles_$n.txt
