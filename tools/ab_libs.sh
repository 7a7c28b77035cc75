#!/bin/bash
# usage (GPU box): tools/ab_libs.sh "<variants>" "<cfgs>" "<quick.py settings>" -- tools/quick.py with the product library ("base") and with each
# mustafar_amd/lib/variants/libmustafar_hip_<name>.so, same box, two rounds alternating.  One line per (library, config, setting).
VARS=${1:-base}; CFGS=${2:-c3}; SETS=${3:-dot2}
for rep in 1 2; do for lib in $VARS; do
  if [ $lib = base ]; then unset MUSTAFAR_HIP_LIB; else export MUSTAFAR_HIP_LIB=$PWD/mustafar_amd/lib/variants/libmustafar_hip_$lib.so; fi
  timeout -k 10 300 python tools/quick.py --cfg $CFGS --set $SETS 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib /" || exit 1
done; done
