#!/bin/bash
# usage (GPU box): tools/ab_variant.sh <variant-name> [cfgs]  -- bench legs (both FMA engines, fused + graph) with the product library and
# with mustafar_amd/lib/variants/libmustafar_hip_<name>.so, same box, alternating
V=$1; CFGS=${2:-"c3 c5"}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-api --no-other-configs --no-trigger-leg"
for rep in 1 2; do for lib in base $V; do
  if [ $lib = base ]; then unset MUSTAFAR_HIP_LIB; else export MUSTAFAR_HIP_LIB=$PWD/mustafar_amd/lib/variants/libmustafar_hip_$lib.so; fi
  for c in $CFGS; do
    python bench.py --config $c $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d.get('fma_engine_mfma') or {}; r=d['roofline']; rm=d.get('roofline_mfma') or {}
print('$lib $c valu', d['value'], r['kernel'][:14], r['avg_launch_us'], r['frac'], '| mfma', m.get('value'), rm.get('avg_launch_us'), rm.get('frac'))"
  done
done; done
