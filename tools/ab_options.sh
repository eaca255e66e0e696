#!/bin/bash
# Same-box A/B of launcher options: tools/ab_options.sh "<opts A>" "<opts B>" [bench args]   (opts like "k32=27" or "k32=59 k32_sb_min_wgs=512")
# Alternates A B A B (two passes each) so that box-to-box and thermal drift cancel; prints images/s per run.
A="$1"; B="$2"; shift 2
for rep in 1 2; do
  for o in "$A" "$B"; do
    args=""; for kv in $o; do args="$args --debug-option $kv"; done
    v=$(python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sub-records $args "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%.2f img/s  conv frac %.4f' % (r['value'], r['roofline']['frac']))")
    echo "[$o] $v"
  done
done
