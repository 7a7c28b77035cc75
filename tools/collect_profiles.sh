#!/bin/bash
# Round evidence in one GPU call: usage (GPU box, repo root): tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/
#   bench line (driver form), rocprofv3 kernel stats of the same command, SQ counters (c5, both FMA engines), TCC traffic (c3),
#   mem_spd harness, prefill-compression and append timings, launch-form sweep
TAG=${1:-r02}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
tools/prof_traffic.sh ${TAG} c3 > $O/traffic_c3_tcc.txt 2>&1; python3 tools/make_traffic_json.py gpurun_out/traffic_${TAG} c3 >> $O/traffic_c3_tcc.txt 2>&1; cp profiles/hbm_traffic.json $O/hbm_traffic.json; echo "traffic done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err && echo "bench done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-api --no-other-configs --no-trigger-leg > $O/bench_c3_under_rocprof.json 2>/dev/null )
cp $(ls $O/rocprof_bench/*/*kernel_stats.csv | head -1) $O/bench_c3_kernel_stats.csv && echo "rocprof done"
tools/prof_pmc.sh ${TAG}_valu c5 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}_valu.txt $O/pmc_sq_c5_valu.txt
PROF_FUSED=1 tools/prof_pmc.sh ${TAG}_valu_c3 c3 > /dev/null 2>&1; for p in A B C; do python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_valu_c3_$p --match onepass_kernel; done > $O/pmc_sq_c3_valu_onepass.txt
PROF_FUSED=1 MUSTAFAR_FMA_ENGINE=mfma tools/prof_pmc.sh ${TAG}_mfma_c3 c3 > /dev/null 2>&1; for p in A B C; do python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_mfma_c3_$p --match onepass_kernel; done > $O/pmc_sq_c3_mfma_onepass.txt
MUSTAFAR_FMA_ENGINE=mfma tools/prof_pmc.sh ${TAG}_mfma c5 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}_mfma.txt $O/pmc_sq_c5_mfma.txt; echo "pmc done"
python3 tools/mem_spd.py --api fused reference > $O/mem_spd.txt 2>&1; python3 tools/mem_spd.py --api fused --graph >> $O/mem_spd.txt 2>&1; echo "mem_spd done"
python3 tools/bench_compress.py c3 c4 2>&1 | grep cfg > $O/compress.txt; python3 tools/bench_append.py 2>&1 | grep cfg > $O/append.txt
python3 tools/microbench.py --cfg c2 c3 c4 c5 --rows 1 --iters 30 2>&1 | grep cfg > $O/microbench_valu.txt
MUSTAFAR_FMA_ENGINE=mfma python3 tools/microbench.py --cfg c3 c4 c5 --rows 1 --iters 30 2>&1 | grep cfg > $O/microbench_mfma.txt
python3 tools/microbench.py --cfg c3 c5 --rows 1 --iters 30 --adversarial 2>&1 | grep cfg > $O/microbench_adversarial.txt
echo "all done"; ls $O
