#!/bin/bash
# Partial evidence (round 4b: the compression kernels changed, spmv.hip did not): usage (GPU box, repo root): tools/collect_prefill.sh <tag>
#   the bench line (with its prefill_compression leg), prefill-compression / append / trigger timings, the mem_spd harness.
set -e -o pipefail
TAG=${1:-r04b}; R=$(pwd); O=$R/gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
nonempty() { [ -s "$1" ] || { echo "empty evidence file $1"; exit 1; }; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; nonempty $O/bench_c3.json; echo "bench done"
python3 tools/bench_compress.py c3 c4 2> $O/compress.err | grep cfg > $O/compress.txt; nonempty $O/compress.txt
python3 tools/bench_append.py 2> $O/append.err | grep cfg > $O/append.txt; nonempty $O/append.txt
python3 tools/bench_extent_append.py 2> $O/extent_append.err > $O/extent_append.txt; nonempty $O/extent_append.txt
python3 tools/mem_spd.py --api fused reference > $O/mem_spd.txt 2> $O/mem_spd.err; python3 tools/mem_spd.py --api fused --graph >> $O/mem_spd.txt 2>> $O/mem_spd.err; echo "mem_spd done"
rm -rf $O/rocprof
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/tools/bench_compress.py c3 > /dev/null 2> $O/rocprof.err)
cp $(ls $O/rocprof/*/*kernel_stats.csv | head -1) $O/compress_kernel_stats.csv; rm -rf $O/rocprof
echo "all done"; ls $O
