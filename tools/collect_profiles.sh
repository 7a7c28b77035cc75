#!/bin/bash
# Round evidence in one GPU call: usage (GPU box, repo root): tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/
#   TCC traffic (c3, c4, c5), bench line (driver form), rocprofv3 kernel stats of the same command, SQ counters of the one-pass launch
#   per engine (c3) and on the default engine at c2 / c4 / c5, wave timelines (c3, c4), mem_spd harness, prefill-compression and
#   append timings, the two reference entry points, launch structures, the cost of a trigger (per layer vs batched).
# Every profiler pass starts from an empty directory and keeps its output in a .err file; a failing step -- a program that dies in
# front of a `| grep` included (pipefail) -- stops the script, so nothing stale or truncated can be copied into profiles/ and a fault
# under the profiler does not go unnoticed.
set -e -o pipefail
TAG=${1:-r04}; R=$(pwd); O=$R/gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
one_csv() { local n; n=$(ls $1 2>/dev/null | wc -l); [ "$n" = "1" ] || { echo "expected exactly one file for $1, found $n"; exit 1; }; ls $1; }
nonempty() { [ -s "$1" ] || { echo "empty evidence file $1"; exit 1; }; }

for C in c3 c4 c5; do
  tools/prof_traffic.sh ${TAG}_$C $C > $O/traffic_${C}_tcc.txt 2> $O/traffic_${C}_tcc.err
  python3 tools/make_traffic_json.py gpurun_out/traffic_${TAG}_$C $C >> $O/traffic_${C}_tcc.txt; nonempty $O/traffic_${C}_tcc.txt
done
cp profiles/hbm_traffic.json $O/hbm_traffic.json; echo "traffic done"

python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; nonempty $O/bench_c3.json; echo "bench done"

rm -rf $O/rocprof_bench
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-api --no-other-configs --no-trigger-leg --no-seq-sweep > $O/bench_c3_under_rocprof.json 2> $O/rocprof_bench.err)
cp $(one_csv "$O/rocprof_bench/*/*kernel_stats.csv") $O/bench_c3_kernel_stats.csv; echo "rocprof done"

for E in dot2 valu mfma; do
  PROF_FUSED=1 MUSTAFAR_FMA_ENGINE=$E tools/prof_pmc.sh ${TAG}_${E}_c3 c3 > /dev/null
  grep -v "spmv_kernel\|^$" gpurun_out/pmc_${TAG}_${E}_c3.txt > $O/pmc_sq_c3_${E}_onepass.txt; nonempty $O/pmc_sq_c3_${E}_onepass.txt
done
for C in c2 c4 c5; do
  PROF_FUSED=1 MUSTAFAR_FMA_ENGINE=dot2 tools/prof_pmc.sh ${TAG}_dot2_$C $C > /dev/null
  cp gpurun_out/pmc_${TAG}_dot2_$C.txt $O/pmc_sq_${C}_dot2.txt; nonempty $O/pmc_sq_${C}_dot2.txt
done
PROF_FUSED=1 MUSTAFAR_FMA_ENGINE=mfma tools/prof_pmc.sh ${TAG}_mfma_c5 c5 > /dev/null; cp gpurun_out/pmc_${TAG}_mfma_c5.txt $O/pmc_sq_c5_mfma.txt
echo "pmc done"

python3 tools/wave_trace_onepass.py --cfg c3 --set dot2 mfma > $O/wave_trace_c3.txt 2> $O/wave_trace.err; nonempty $O/wave_trace_c3.txt
python3 tools/wave_trace_onepass.py --cfg c4 --set dot2 mfma > $O/wave_trace_c4.txt 2>> $O/wave_trace.err; nonempty $O/wave_trace_c4.txt
python3 tools/wave_trace_onepass.py --cfg c5 --set dot2 > $O/wave_trace_c5.txt 2>> $O/wave_trace.err; echo "wave traces done"

python3 tools/mem_spd.py --api fused reference > $O/mem_spd.txt 2> $O/mem_spd.err; python3 tools/mem_spd.py --api fused --graph >> $O/mem_spd.txt 2>> $O/mem_spd.err; echo "mem_spd done"
python3 tools/bench_compress.py c3 c4 2> $O/compress.err | grep cfg > $O/compress.txt
python3 tools/bench_append.py 2> $O/append.err | grep cfg > $O/append.txt
MUSTAFAR_FMA_ENGINE=valu python3 tools/microbench.py --cfg c3 c2 c3 c4 c5 --rows 1 8 --iters 30 2> $O/microbench.err | grep cfg > $O/microbench_valu.txt
MUSTAFAR_FMA_ENGINE=mfma python3 tools/microbench.py --cfg c3 c3 c4 c5 --rows 1 --iters 30 2>> $O/microbench.err | grep cfg > $O/microbench_mfma.txt
python3 tools/quick.py --cfg c2 c3 c4 c5 --set dot2 valu mfma valu:onepass=0 valu:lean=0 dot2:tbw=2 mfma:tbw=2 2> $O/quick.err | grep cfg > $O/structures.txt
python3 tools/quick.py --cfg m8 g2 t8192 t8448 --set dot2 valu:onepass=0 2>> $O/quick.err | grep cfg >> $O/structures.txt
python3 tools/bench_extent_append.py 2> $O/extent_append.err > $O/extent_append.txt; nonempty $O/extent_append.txt
# only the summaries travel back (gpurun merges at most 64 MiB): the profiler's raw directories are dropped
rm -rf $R/gpurun_out/pmc_${TAG}_* $R/gpurun_out/traffic_${TAG}_* $O/rocprof_bench
echo "all done"; ls $O
