# f16-mode evidence (round 6): kernel stats of the B = 64 graph loop from the same bench.py command, beside the bf16 one; then the traffic passes and a driver line.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
H="--no-cpu-baseline --no-sub-records"
for n in f16 bf16; do
  rm -rf $O/stats_${n}_b64
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${n}_b64 -o s -- python3 $R/bench.py $H --precision $n --batch 64 --graph --steps 4 --warmup 1 > $O/stats_${n}_b64.log 2>&1 < /dev/null
  f=$(find $O/stats_${n}_b64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_${n}_b64_graph_f16tree.csv
  grep -h '"metric"' $O/stats_${n}_b64.log | tail -1 > $O/bench_under_rocprof_${n}_b64_graph_f16tree.json
  rm -rf $O/stats_${n}_b64
done
cd $R
bash tools/refresh_traffic_r06.sh > $O/traffic_refresh.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_f16tree.json 2> $O/bench_driver_cmd_f16tree.err
echo f16-refresh-done
