#!/bin/bash
# copy what tools/refresh_profiles_r06.sh left in gpurun_out/r06 into profiles/ under the names DESIGN.md §6 cites
set -e
cd $(dirname $0)/..
O=gpurun_out/r06; P=profiles
for n in f16x3_b16 f32_b16 bf16_b64_graph f16x3_b1_graph f16x3_b1_512_graph f16x3_b16_128; do cp $O/kernel_stats_$n.csv $P/r06_kernel_stats_$n.csv; cp $O/bench_under_rocprof_$n.json $P/r06_bench_under_rocprof_$n.json; done
cp $O/kernel_stats_train.csv $P/r06_kernel_stats_train_f16x3_b32.csv; cp $O/kernel_stats_train32.csv $P/r06_kernel_stats_train_f32_b32.csv
cp $O/bench_train_f16x3_b32.json $P/r06_bench_train_f16x3_b32.json; cp $O/bench_train_f32_b32.json $P/r06_bench_train_f32_b32.json
for n in f16x3_b16 bf16_b64; do cp $O/pmc_mfma_util_$n.txt $P/r06_pmc_mfma_util_$n.txt; cp $O/pmc_wave_cycles_$n.txt $P/r06_pmc_wave_cycles_$n.txt; done
cp $O/pmc_mfma_util_train_f16x3_b32.txt $P/r06_pmc_mfma_util_train_f16x3_b32.txt
cp $O/pmc_hbm_traffic_f16x3_b16.json $P/r06_pmc_hbm_traffic_f16x3_b16.json; cp $O/pmc_hbm_traffic_bf16_b64.json $P/r06_pmc_hbm_traffic_bf16_b64.json
cp $O/bench_driver_cmd.json $P/r06_bench_driver_cmd_profiles_box.json
