#!/bin/bash
# usage (GPU box): tools/prof_cfg.sh <cfg> [env...]  -- rocprofv3 kernel stats of the fused + graph leg of one config (default engine)
C=$1; R=$(pwd); O=$R/gpurun_out/prof_$C; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-reference-api --no-other-configs --no-trigger-leg > $O/bench.json 2>/dev/null
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("$O/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(8), r["Percentage"].rjust(6))
PY
