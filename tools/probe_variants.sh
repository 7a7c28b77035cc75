#!/bin/bash
# usage (GPU box): tools/probe_variants.sh <out-file> <cfg> "<settings>" variant...   -- quick.py under each probe build (timing only)
OUT=$1; CFG=$2; SETS=$3; shift 3
for v in "$@"; do
  if [ "$v" = "asbuilt" ]; then LIB=""; else LIB="mustafar_amd/lib/variants/libmustafar_hip_$v.so"; fi
  echo "## variant $v" >> $OUT
  MUSTAFAR_HIP_LIB=$LIB timeout -k 10 300 python tools/quick.py --cfg $CFG --steps 10 --set $SETS 2>&1 | grep '"cfg"' >> $OUT
done
