#!/bin/bash
# usage (GPU box): tools/prof_cfgs.sh "<cfgs>" [extra bench args] -- bench.py's headline leg of each config under rocprofv3 --kernel-trace --stats:
# the replayed graph's own kernel durations (what the eager pass behind the replays cannot give).  One block per config on stdout.
R=$(pwd)
for C in $1; do
  D=$R/gpurun_out/prof_cfg_$C; rm -rf $D; mkdir -p $D
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-reference-api --no-other-configs --no-trigger-leg --no-seq-sweep ${@:2} > $D.json 2> $D.err) || { echo "$C: failed"; tail -3 $D.err; exit 1; }
  echo "== $C  $(python3 -c "import json;d=json.loads(open('$D.json').readline());print('tokens/s',d['value'],'ms/step',d['ms_per_step'],'eager kernel us',d['roofline']['avg_launch_us'])")"
  python3 tools/kstats.py $D 6
done
