#!/bin/bash
# HBM traffic of the decode kernels from the TCC counters, one counter per pass (they do not fit together).
# usage (GPU box, repo root): tools/prof_traffic.sh <tag> [cfg]  -> gpurun_out/traffic_<tag>.txt
# Each pass starts from an empty directory, keeps the profiler's output in <dir>.err and stops the script on failure.
set -e
TAG=$1; CFGN=${2:-c3}; R=$(pwd)
export PROF_FUSED=1
for P in "F FETCH_SIZE" "W WRITE_SIZE"; do
  set -- $P
  D=$R/gpurun_out/traffic_${TAG}_$1
  rm -rf "$D"; mkdir -p "$D"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$D" -- python3 $R/tools/prof_driver.py $CFGN 6 > "$D.err" 2>&1) \
    || { echo "rocprofv3 $2 pass failed: see $D.err"; tail -5 "$D.err"; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/traffic_${TAG}_F > gpurun_out/traffic_${TAG}.txt
python3 tools/pmc_summary.py gpurun_out/traffic_${TAG}_W >> gpurun_out/traffic_${TAG}.txt
grep -E "spmv|onepass|combine|vectorized_elementwise|FETCH|WRITE" gpurun_out/traffic_${TAG}.txt | grep -B1 -E "FETCH|WRITE" | grep -v "^--"
