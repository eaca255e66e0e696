# Round-6 evidence, collected on the GPU box into gpurun_out/r06 (copy what is to be judged into profiles/):
#   gpurun --timeout 3000 -- 'bash tools/refresh_profiles_r06.sh'
# Every figure of the driver's bench line gets a rocprofv3 file made by the SAME bench.py command (program directly after `--`),
# so each `frac` can be recomputed from profiles/: kernel stats for the headline (f16x3 B=16) and for every sub-record regime
# (exact f32 B=16, bf16 B=64 + hipGraph, f16x3 B=1 + hipGraph, both training steps), PMC utilisation + a wave-cycle breakdown for
# f16x3 B=16 and bf16 B=64, and the HBM traffic passes.  Counters are collected in runs of their own (no trace domains but
# --kernel-trace).  Every step is bounded by `timeout` and reads nothing from stdin.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 < /dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err < /dev/null
timeout 600 python bench.py --train --precision f16x3 --steps 3 --warmup 1 > $O/bench_train_f16x3_b32.json 2>/dev/null < /dev/null
timeout 600 python bench.py --train --precision f32 --steps 3 --warmup 1 > $O/bench_train_f32_b32.json 2>/dev/null < /dev/null
cd /tmp && export TMPDIR=/tmp
H="--no-cpu-baseline --no-sub-records"
stats() {   # name, bench args...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -o s -- python3 $R/bench.py $H "$@" > $O/stats_$n.log 2>&1 < /dev/null
  f=$(find $O/stats_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$n.csv
  grep -h '"metric"' $O/stats_$n.log | tail -1 > $O/bench_under_rocprof_$n.json
}
stats f16x3_b16 --steps 20 --warmup 5
stats f32_b16 --precision f32 --steps 4 --warmup 1
stats bf16_b64_graph --precision bf16 --batch 64 --graph --steps 4 --warmup 1
stats f16x3_b1_graph --batch 1 --graph --steps 20 --warmup 3
# kernel choice outside the tuned 256 x 256 shapes: infer.py's 512 x 512 B = 1 and a 128 x 128 B = 16 batch (which kernels the launch rules pick there)
stats f16x3_b1_512_graph --size 512 --batch 1 --graph --steps 3 --warmup 1
stats f16x3_b16_128 --size 128 --batch 16 --steps 3 --warmup 1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -o t -- python3 $R/bench.py --train --precision f16x3 --steps 2 --warmup 1 > $O/stats_train.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train32 -o t -- python3 $R/bench.py --train --precision f32 --steps 2 --warmup 1 > $O/stats_train32.log 2>&1 < /dev/null
for d in stats_train stats_train32; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_$d.csv; done
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/pmc_fetch.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/pmc_write.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_bf16 -o f -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile --precision bf16 --batch 64 > $O/pmc_fetch_bf16.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_bf16 -o w -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile --precision bf16 --batch 64 > $O/pmc_write_bf16.log 2>&1 < /dev/null
PA="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM"
PB="GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
PC="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
util() {    # name, bench args...
  n=$1; shift
  timeout 600 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/util_a_$n -o a -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile "$@" > $O/util_a_$n.log 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/util_b_$n -o b -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile "$@" > $O/util_b_$n.log 2>&1 < /dev/null
  timeout 600 rocprofv3 --pmc $PC --kernel-trace --output-format csv -d $O/util_c_$n -o c -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile "$@" > $O/util_c_$n.log 2>&1 < /dev/null
}
util f16x3_b16
util bf16_b64 --precision bf16 --batch 64
T="--train --precision f16x3"
timeout 600 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/tutil_a -o a -- python3 $R/bench.py $T --steps 1 --warmup 0 > $O/tutil_a.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/tutil_b -o b -- python3 $R/bench.py $T --steps 1 --warmup 0 > $O/tutil_b.log 2>&1 < /dev/null
cd $R
python tools/pmc_util.py $O/tutil_a $O/tutil_b 24 > $O/pmc_mfma_util_train_f16x3_b32.txt 2> $O/pmc_tutil.err < /dev/null
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_hbm_traffic_f16x3_b16.json 2> $O/pmc_traffic.err < /dev/null
python tools/pmc_traffic.py $O/pmc_fetch_bf16 $O/pmc_write_bf16 > $O/pmc_hbm_traffic_bf16_b64.json 2> $O/pmc_traffic_bf16.err < /dev/null
for n in f16x3_b16 bf16_b64; do
  python tools/pmc_util.py $O/util_a_$n $O/util_b_$n 16 > $O/pmc_mfma_util_$n.txt 2> $O/pmc_util_$n.err < /dev/null
  python tools/pmc_wave.py $O/util_c_$n $O/util_a_$n 12 > $O/pmc_wave_cycles_$n.txt 2> $O/pmc_wave_$n.err < /dev/null
done
rm -rf $O/pmc_fetch/*/ $O/pmc_write/*/ $O/pmc_fetch_bf16/*/ $O/pmc_write_bf16/*/ 2>/dev/null
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
ls -la $O
echo refresh-done
