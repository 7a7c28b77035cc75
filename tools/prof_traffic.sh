#!/bin/bash
# HBM traffic of the SpMV kernels from the TCC counters, one counter per pass (they do not fit together).
# usage (GPU box, repo root): tools/prof_traffic.sh <tag> [cfg]  -> gpurun_out/traffic_<tag>.txt
TAG=$1; CFGN=${2:-c3}; R=$(pwd)
cd /tmp && export TMPDIR=/tmp && export PROF_FUSED=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_${TAG}_F -- python3 $R/tools/prof_driver.py $CFGN 6 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_${TAG}_W -- python3 $R/tools/prof_driver.py $CFGN 6 > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/traffic_${TAG}_F > gpurun_out/traffic_${TAG}.txt
python3 tools/pmc_summary.py gpurun_out/traffic_${TAG}_W >> gpurun_out/traffic_${TAG}.txt
grep -E "spmv|onepass|combine|vectorized_elementwise|FETCH|WRITE" gpurun_out/traffic_${TAG}.txt | grep -B1 -E "FETCH|WRITE" | grep -v "^--"
