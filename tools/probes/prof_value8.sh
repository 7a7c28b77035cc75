#!/bin/bash
# usage (GPU box): tools/probes/prof_value8.sh ["<variant names>"] -- the two entry points at c3 with 8 rows (the hook's padded operands) under rocprofv3
# --kernel-trace --stats: the kernels' own durations, for the product library with mustafar_tune(13, 1 | 0) and for each named variant library.
R=$(pwd)
run() {   # $1 = label, $2 = tune value
  D=$R/gpurun_out/prof_value8_$1; rm -rf $D; mkdir -p $D
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/microbench.py --cfg c3 --rows 8 --iters 50 --tune 13=$2 > $D.json 2> $D.err) || { echo "failed"; tail -3 $D.err; exit 1; }
  echo "== $1 (tune 13=$2)  $(grep -v amdgpu $D.json | cut -c1-120)"
  python3 tools/kstats.py $D 8 | grep -v "at::native\|tile_\|prune_"
  rm -rf $D
}
unset MUSTAFAR_HIP_LIB
run product 1 && run product_round1_kernel 0
for V in $1; do export MUSTAFAR_HIP_LIB=$R/mustafar_amd/lib/variants/libmustafar_hip_$V.so; run $V 1; done
