#!/bin/bash
# like tools/ab_libs.sh but goes on behind a failed self-check (timing-only probe builds give wrong results by design)
VARS=${1:-base}; CFGS=${2:-c3}; SETS=${3:-dot2}
for rep in 1 2; do for lib in $VARS; do
  if [ $lib = base ]; then unset MUSTAFAR_HIP_LIB; else export MUSTAFAR_HIP_LIB=$PWD/mustafar_amd/lib/variants/libmustafar_hip_$lib.so; fi
  timeout -k 10 300 python tools/quick.py --cfg $CFGS --set $SETS 2>&1 | grep "^{" | cut -c1-230 | sed "s/^/$lib /"
done; done
