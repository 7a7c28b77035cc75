#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_pmc.sh <tag> [cfg]   -> gpurun_out/pmc_<tag>.txt
# (SQ counters of the two SpMV kernels, three --pmc passes of <= 8 SQ counters each; engine / form via the environment)
TAG=$1; CFGN=${2:-c5}; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_A -- python3 $R/tools/prof_driver.py $CFGN 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_SMEM SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_B -- python3 $R/tools/prof_driver.py $CFGN 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_C -- python3 $R/tools/prof_driver.py $CFGN 6 > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_A --match "spmv_kernel|onepass_kernel" > gpurun_out/pmc_${TAG}.txt
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_B --match "spmv_kernel|onepass_kernel" >> gpurun_out/pmc_${TAG}.txt
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_C --match "spmv_kernel|onepass_kernel" >> gpurun_out/pmc_${TAG}.txt
cat gpurun_out/pmc_${TAG}.txt
