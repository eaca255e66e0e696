#!/bin/bash
# MFMA busy / effective clock / instruction mix per kernel under several library builds (one rocprofv3 --pmc pass each over tools/lib_probe.py):
#   bash tools/pmc_variants.sh <prec> <batch> <rows> <label>=<lib.so or ''>[:opt=val,...] ...
# The effective clock (GRBM_GUI_ACTIVE / 8 / time) tells a knock-out build's gain from removed stalls apart from a gain from the
# higher clock the chip holds on cheaper data (knock-outs change the operands: MI355X_MICROARCH.md, DVFS give-back).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcv; mkdir -p $O
PREC=$1; B=$2; ROWS=$3; shift 3
PA="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM"
PB="GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}; lib=${rest%%:*}; opts=""
  if [[ "$rest" == *:* ]]; then opts=${rest#*:}; fi
  if [ -n "$lib" ]; then export FDSR_LIB=$R/$lib; else unset FDSR_LIB; fi
  export PROBE_OPTS="$opts" REPS=1
  rm -rf $O/a_$label $O/b_$label
  timeout 600 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/a_$label -o a -- python3 $R/tools/lib_probe.py $PREC $B > $O/a_$label.log 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/b_$label -o b -- python3 $R/tools/lib_probe.py $PREC $B > $O/b_$label.log 2>&1 < /dev/null
  echo "== $label ($lib $opts)"
  python3 $R/tools/pmc_util.py $O/a_$label $O/b_$label $ROWS
  rm -rf $O/a_$label $O/b_$label
done
