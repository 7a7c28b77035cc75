#!/bin/bash
# usage (GPU box): tools/probes/ab_value8_variants.sh "<variant names>" -- the two entry points with 8 rows at c3 / c4, the product library and each variant under
# mustafar_amd/lib/variants/, twice, interleaved
for rep in 1 2; do
for V in "" $1; do
  if [ -n "$V" ]; then export MUSTAFAR_HIP_LIB=$PWD/mustafar_amd/lib/variants/libmustafar_hip_$V.so; else unset MUSTAFAR_HIP_LIB; fi
  echo "== ${V:-product}"
  python tools/microbench.py --cfg c2 c3 c4 --rows 1 8 --iters 50 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['cfg'], 'rows', d['rows'], 'key', d['key_us'], 'value', d['value_us'])"
done
done
