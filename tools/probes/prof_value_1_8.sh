R=$(pwd); D=$R/gpurun_out/prof_v18; rm -rf $D; mkdir -p $D
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/microbench.py --cfg c3 --rows 1 8 --iters 50 > $D.json 2> $D.err)
grep -v amdgpu $D.json | cut -c1-130; python3 tools/kstats.py $D 10 | grep -v "at::native\|tile_\|prune_" | cut -c1-130; rm -rf $D
