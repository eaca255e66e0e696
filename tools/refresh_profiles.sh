set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
timeout 400 python bench.py > $O/bench_f16x3_b16.json 2> $O/bench_f16x3_b16.err
timeout 300 python bench.py --precision f32 --no-cpu-baseline > $O/bench_f32_b16.json 2>/dev/null
timeout 300 python bench.py --precision bf16 --no-cpu-baseline > $O/bench_bf16_b16.json 2>/dev/null
timeout 300 python bench.py --batch 64 --graph --steps 2 --no-cpu-baseline > $O/bench_f16x3_b64_graph.json 2>/dev/null
timeout 300 python bench.py --precision bf16 --batch 64 --graph --steps 2 --no-cpu-baseline > $O/bench_bf16_b64_graph.json 2>/dev/null
timeout 300 python bench.py --batch 1 --graph --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_f16x3_b1_graph.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b16 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $O/pmc_write.log 2>&1
# MFMA / VALU / LDS utilisation: two more counter passes (<= 8 SQ counters each), summarised by tools/pmc_util.py
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/util_a -o a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $O/util_a.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/util_b -o b -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $O/util_b.log 2>&1
ls -la $O $O/stats $O/pmc_fetch | head -40
