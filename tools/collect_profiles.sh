#!/bin/bash
# Round evidence in one GPU call: usage (GPU box, repo root): tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/
#   TCC traffic (c3 itemised with the no-stream probe build, c2, b1, c4, c5), bench line (driver form), rocprofv3 kernel stats of the same command, SQ counters of the one-pass launch
#   per engine (c3) and on the default engine at c2 / c4 / c5, wave timelines (c3, c4), mem_spd harness, prefill-compression and
#   append timings, the two reference entry points, launch structures, the cost of a trigger (per layer vs batched).
# Every profiler pass starts from an empty directory and keeps its output in a .err file; a failing step -- a program that dies in
# front of a `| grep` included (pipefail) -- stops the script, so nothing stale or truncated can be copied into profiles/ and a fault
# under the profiler does not go unnoticed.
set -e -o pipefail
TAG=${1:-r06}; PART=${2:-all}; R=$(pwd); O=$R/gpurun_out/$TAG
# PART: A = traffic, bench line, rocprofv3 kernel stats, SQ counters; B = the rest; all = both (a call of gpurun is limited to 20 minutes:
# `tools/collect_profiles.sh r06 A` and `... r06 B` as two calls)
if [ "$PART" != B ]; then rm -rf "$O"; fi; mkdir -p "$O"
one_csv() { local n; n=$(ls $1 2>/dev/null | wc -l); [ "$n" = "1" ] || { echo "expected exactly one file for $1, found $n"; exit 1; }; ls $1; }
nonempty() { [ -s "$1" ] || { echo "empty evidence file $1"; exit 1; }; }

if [ "$PART" != B ]; then
# round 6: c3 / c2 / b1 through tools/traffic_items.sh (the product library + the no-stream probe build at c3: itemised traffic), then c4 / c5
tools/traffic_items.sh $TAG > $O/traffic_items.txt 2> $O/traffic_items.err; nonempty $O/traffic_items.txt
for C in c4 c5; do
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
for C in c2 c4 c5 b1; do
  PROF_FUSED=1 MUSTAFAR_FMA_ENGINE=dot2 tools/prof_pmc.sh ${TAG}_dot2_$C $C > /dev/null
  cp gpurun_out/pmc_${TAG}_dot2_$C.txt $O/pmc_sq_${C}_dot2.txt; nonempty $O/pmc_sq_${C}_dot2.txt
done
PROF_FUSED=1 MUSTAFAR_FMA_ENGINE=mfma tools/prof_pmc.sh ${TAG}_mfma_c5 c5 > /dev/null; cp gpurun_out/pmc_${TAG}_mfma_c5.txt $O/pmc_sq_c5_mfma.txt
echo "pmc done"
fi
if [ "$PART" = A ]; then   # (only the summaries travel back: gpurun merges at most 64 MiB)
  rm -rf $R/gpurun_out/pmc_${TAG}_* $R/gpurun_out/traffic_${TAG}_* $O/rocprof_bench
  echo "part A done"; ls $O; exit 0
fi

python3 tools/wave_trace_onepass.py --cfg c3 --set dot2 mfma > $O/wave_trace_c3.txt 2> $O/wave_trace.err; nonempty $O/wave_trace_c3.txt
python3 tools/wave_trace_onepass.py --cfg c4 --set dot2 mfma > $O/wave_trace_c4.txt 2>> $O/wave_trace.err; nonempty $O/wave_trace_c4.txt
python3 tools/wave_trace_onepass.py --cfg c5 --set dot2 > $O/wave_trace_c5.txt 2>> $O/wave_trace.err
python3 tools/wave_trace_onepass.py --cfg c2 --set valu > $O/wave_trace_c2.txt 2>> $O/wave_trace.err; nonempty $O/wave_trace_c2.txt
python3 tools/wave_trace_onepass.py --cfg b1 --set dot2 > $O/wave_trace_b1.txt 2>> $O/wave_trace.err; nonempty $O/wave_trace_b1.txt; echo "wave traces done"
# round 6: the in-kernel clock 8 ms ... 580 ms into a run, and the product library's ms/step over a 400-replay run
python3 tools/clock_probe.py > $O/clocks_raw.jsonl 2> $O/clocks.err; nonempty $O/clocks_raw.jsonl; echo "clock probe done"

python3 tools/mem_spd.py --api fused reference > $O/mem_spd.txt 2> $O/mem_spd.err; python3 tools/mem_spd.py --api fused --graph >> $O/mem_spd.txt 2>> $O/mem_spd.err; echo "mem_spd done"
python3 tools/bench_compress.py c3 c4 2> $O/compress.err | grep cfg > $O/compress.txt
python3 tools/probes/convert_breakdown.py c1 c2 c3 c4 c5 2> $O/convert_breakdown.err | grep cfg > $O/convert_breakdown.txt; nonempty $O/convert_breakdown.txt
tools/probes/prof_value8.sh > $O/value8_kernels.txt 2> $O/value8_kernels.err; nonempty $O/value8_kernels.txt
hipcc --offload-arch=gfx950 -O2 -o tools/ubench/inlaunch_merge tools/ubench/inlaunch_merge.hip 2> $O/inlaunch_merge.err && timeout -k 10 120 ./tools/ubench/inlaunch_merge > $O/inlaunch_merge.txt 2>> $O/inlaunch_merge.err; nonempty $O/inlaunch_merge.txt
python3 tools/bench_append.py 2> $O/append.err | grep cfg > $O/append.txt
MUSTAFAR_FMA_ENGINE=valu python3 tools/microbench.py --cfg c3 c2 c3 c4 c5 --rows 1 8 --iters 30 2> $O/microbench.err | grep cfg > $O/microbench_valu.txt
MUSTAFAR_FMA_ENGINE=mfma python3 tools/microbench.py --cfg c3 c3 c4 c5 --rows 1 --iters 30 2>> $O/microbench.err | grep cfg > $O/microbench_mfma.txt
python3 tools/quick.py --cfg c2 c3 c4 c5 --set dot2 valu mfma dot2:sb=0 valu:sb=0 mfma:sb=0 valu:onepass=0 dot2:tbw=1 2> $O/quick.err | grep cfg > $O/structures.txt
python3 tools/quick.py --cfg m8 g2 b1 --set dot2 valu:onepass=0 2>> $O/quick.err | grep cfg >> $O/structures.txt
# round 6: the small-launch kernel against the super-block kernel, forced both ways, from below one wave per SIMD (c1, 4k x batch 1) to 3.75 (c2)
python3 tools/quick.py --cfg c1 b1s b1 c2 --set dot2:small=2 dot2:small=0 valu:small=2 valu:small=0 2>> $O/quick.err | grep cfg > $O/small_launch.txt; nonempty $O/small_launch.txt
# round 5: off the grid of whole rounds of workgroups (super-block form with / without the raised priority of a small last round, round 4's pair form)
python3 tools/quick.py --cfg t8192 t8448 t8704 t8960 t9216 t10240 --set dot2 dot2:late=0 dot2:sb=0 2>> $O/quick.err | grep cfg > $O/offgrid.txt; nonempty $O/offgrid.txt
tools/prof_cfgs.sh "c3 t8192 t8448 t8704" > $O/offgrid_kernel_stats.txt 2> $O/offgrid_kernel_stats.err; nonempty $O/offgrid_kernel_stats.txt
python3 tools/bench_extent_append.py 2> $O/extent_append.err > $O/extent_append.txt; nonempty $O/extent_append.txt
# round 5: the instruction classes of the timed kernel's ISA (static; tools/isa_breakdown.py) next to the counters
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -S --cuda-device-only -o /tmp/spmv_isa.s mustafar_amd/csrc/spmv.hip 2> /dev/null
{ echo "== decode_onepass_sb_kernel<2, false, 4, false> (dot2; the timed kernel): text between the trip markers holds the two-block pipelines AND the one-block alternatives: 384 tiles of text for the 256 a wave walks"; python3 tools/isa_breakdown.py /tmp/spmv_isa.s 'decode_onepass_sb_kernelILi2ELb0ELi4ELb0EE' --tiles 384 --markers;
  echo; echo "== decode_onepass_leanpair_kernel<2, false, 4> (round 4's kernel: one trip of its block loop, the even wave's path, 128 tiles)"; python3 tools/isa_breakdown.py /tmp/spmv_isa.s 'decode_onepass_leanpair_kernelILi2ELb0ELi4E'; } > $O/isa_breakdown.txt; nonempty $O/isa_breakdown.txt
# only the summaries travel back (gpurun merges at most 64 MiB): the profiler's raw directories are dropped
rm -rf $R/gpurun_out/pmc_${TAG}_* $R/gpurun_out/traffic_${TAG}_* $O/rocprof_bench
echo "all done"; ls $O
