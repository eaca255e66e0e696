# The HBM-traffic passes of tools/refresh_profiles_r06.sh alone (PMC FETCH_SIZE / WRITE_SIZE in runs of their own), for a tree whose
# sampling-kernel sources changed after the full refresh: bench.py quotes profiles/r06_pmc_hbm_traffic_*.json only while their
# kernel_source_hash matches.   gpurun --timeout 1500 -- 'bash tools/refresh_traffic_r06.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 < /dev/null
cd /tmp && export TMPDIR=/tmp
H="--no-cpu-baseline --no-sub-records"
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_fetch_bf16 $O/pmc_write_bf16
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/pmc_fetch.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile > $O/pmc_write.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_bf16 -o f -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile --precision bf16 --batch 64 > $O/pmc_fetch_bf16.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_bf16 -o w -- python3 $R/bench.py --steps 1 --warmup 0 $H --no-profile --precision bf16 --batch 64 > $O/pmc_write_bf16.log 2>&1 < /dev/null
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_hbm_traffic_f16x3_b16.json 2> $O/pmc_traffic.err < /dev/null
python tools/pmc_traffic.py $O/pmc_fetch_bf16 $O/pmc_write_bf16 > $O/pmc_hbm_traffic_bf16_b64.json 2> $O/pmc_traffic_bf16.err < /dev/null
rm -rf $O/pmc_fetch/*/ $O/pmc_write/*/ $O/pmc_fetch_bf16/*/ $O/pmc_write_bf16/*/ 2>/dev/null
head -c 600 $O/pmc_hbm_traffic_f16x3_b16.json; echo; head -c 600 $O/pmc_hbm_traffic_bf16_b64.json; echo traffic-done
