#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_pmc.sh <tag> [cfg]   -> gpurun_out/pmc_<tag>.txt
# (SQ counters of the SpMV / one-pass kernels, three --pmc passes of <= 8 SQ counters each; engine / form via the environment.
# Each pass starts from an empty output directory, keeps the profiler's own output in <dir>.err and stops the script if the
# profiled program fails: a stale CSV can never stand in for a run that did not happen.)
set -e
TAG=$1; CFGN=${2:-c5}; R=$(pwd)
pass() {   # pass <suffix> <counters...>
  local D=$R/gpurun_out/pmc_${TAG}_$1; shift
  rm -rf "$D"; mkdir -p "$D"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D" -- python3 $R/tools/prof_driver.py $CFGN 6 > "$D.err" 2>&1) \
    || { echo "rocprofv3 pass failed: see $D.err"; tail -5 "$D.err"; exit 1; }
}
pass A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass B SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_SMEM SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY
pass C SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC
: > gpurun_out/pmc_${TAG}.txt
for p in A B C; do python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_$p --match "spmv_kernel|onepass" >> gpurun_out/pmc_${TAG}.txt; done
cat gpurun_out/pmc_${TAG}.txt
