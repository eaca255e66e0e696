# After a kernel-source edit that leaves the conv kernels' behaviour alone: re-collect only what is gated on the source hash
# (the HBM-traffic PMC passes) and the driver-command bench line, into gpurun_out/r03.
#   gpurun --timeout 1500 -- 'bash tools/refresh_traffic_r03.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 < /dev/null
cd /tmp && export TMPDIR=/tmp
H="--no-cpu-baseline --no-sub-records"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/pmc_fetch.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/pmc_write.log 2>&1 < /dev/null
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_hbm_traffic_f16x3_b16.json 2> $O/pmc_traffic.err < /dev/null
rm -rf $O/pmc_fetch/*/ $O/pmc_write/*/ 2>/dev/null
mkdir -p profiles_tmp && cp $O/pmc_hbm_traffic_f16x3_b16.json profiles/r03_pmc_hbm_traffic_f16x3_b16.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err < /dev/null
rmdir profiles_tmp
cat $O/pmc_hbm_traffic_f16x3_b16.json | head -c 600; echo; tail -1 $O/bench_driver_cmd.json | head -c 1500
echo refresh-traffic-done
