#!/bin/bash
# usage (GPU box): tools/quick_bench.sh [cfgs]  -- fused + graph legs of both FMA engines, one line per config
CFGS=${1:-"c3 c5"}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-api --no-other-configs --no-trigger-leg"
for c in $CFGS; do
  python bench.py --config $c $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d.get('fma_engine_mfma') or {}; r=d['roofline']; rm=d.get('roofline_mfma') or {}
print('$c valu', d['value'], d['ms_per_step'], r['kernel'][:14], r['avg_launch_us'], r['frac'], '| mfma', m.get('value'), m.get('ms_per_step'), rm.get('avg_launch_us'), rm.get('frac'))"
done
