#!/bin/bash
# usage (GPU box, repo root): tools/prof_pmc_compress.sh <tag> [cfg]  -> gpurun_out/pmc_compress_<tag>.txt
# SQ counters of the prefill kernels (compress_block_kernel, prune_magnitude_kernel), two --pmc passes; see tools/prof_pmc.sh.
set -e
TAG=$1; CFGN=${2:-c3}; R=$(pwd)
pass() {
  local D=$R/gpurun_out/pmc_compress_${TAG}_$1; shift
  rm -rf "$D"; mkdir -p "$D"
  (cd /tmp && TMPDIR=/tmp FUSED_ONLY=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D" -- python3 $R/tools/bench_compress.py $CFGN > "$D.err" 2>&1) \
    || { echo "rocprofv3 pass failed: see $D.err"; tail -5 "$D.err"; exit 1; }
}
pass A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass B SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM
: > gpurun_out/pmc_compress_${TAG}.txt
for p in A B; do python3 tools/pmc_summary.py gpurun_out/pmc_compress_${TAG}_$p --match "compress_block|prune_magnitude" >> gpurun_out/pmc_compress_${TAG}.txt; done
rm -rf gpurun_out/pmc_compress_${TAG}_A gpurun_out/pmc_compress_${TAG}_B
cat gpurun_out/pmc_compress_${TAG}.txt
