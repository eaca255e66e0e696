# Power / clock of the GPU while the sampling loop runs (f16x3 B=16, then bf16 B=64, then zeros): rocm-smi sampled twice a second.
# gpurun -- 'bash tools/power_probe.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/power; mkdir -p $O; cd $R
rocm-smi --showpower --showclocks --showperflevel --showmaxpower > $O/idle.txt 2>&1
sample() {  # $1 = tag, rest = command
  tag=$1; shift
  "$@" > $O/$tag.out 2>&1 < /dev/null &
  pid=$!
  sleep 6                                 # model build + warm-up
  for i in $(seq 1 12); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' ' >> $O/$tag.smi; echo >> $O/$tag.smi
    sleep 0.5
  done
  wait $pid
}
sample f16x3 timeout 300 python tools/lib_probe.py f16x3 16
sample bf16 timeout 300 python tools/lib_probe.py bf16 64
sample zeros timeout 300 python tools/zero_data_probe.py
for t in f16x3 bf16 zeros; do echo "== $t"; cat $O/$t.smi | cut -c1-220 | head -12; tail -3 $O/$t.out; done
cat $O/idle.txt | head -30
