# MFMA busy / VALU per MFMA / clock per kernel of the f16 mode's B = 64 loop beside bf16's on the same box (PMC passes of their own, as tools/refresh_profiles_r06.sh)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
H="--no-cpu-baseline --no-sub-records"
PA="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM"
PB="GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
for n in f16 bf16; do
  rm -rf $O/futil_a_$n $O/futil_b_$n
  timeout 600 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/futil_a_$n -o a -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile --precision $n --batch 64 > $O/futil_a_$n.log 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/futil_b_$n -o b -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile --precision $n --batch 64 > $O/futil_b_$n.log 2>&1 < /dev/null
  cd $R; python tools/pmc_util.py $O/futil_a_$n $O/futil_b_$n 16 > $O/pmc_mfma_util_${n}_b64_f16tree.txt 2> $O/pmc_futil_$n.err < /dev/null; cd /tmp
  rm -rf $O/futil_a_$n/*/ $O/futil_b_$n/*/
done
head -20 $O/pmc_mfma_util_f16_b64_f16tree.txt
