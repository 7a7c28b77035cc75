#!/usr/bin/env python3
"""A decode leg of `steps` graph-replayed steps through its 256-token triggers (bench.py's timed_graph): tokens/s incl. the triggers, the
one-pass kernel's duration at the END of the leg (the extents instantiation, at T + 256 x triggers) and every trigger step's wall time.

    python tools/bench_trigger_leg.py [c3] [300]
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mustafar_amd import _lib, mustafar_package as mp
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
lib = _lib.load(); timer = bench.KernelTimer(mp); timer.install()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
w = bench.Workload(cfg, 32, dev, 0, 1, None, False, timer, lib)
dt, (ku, vu, n) = w.timed_graph(steps, 1)
print(json.dumps({"cfg": cfg, "steps": steps, "tok_s_incl_triggers": round(w.batch * steps / dt, 1), "kernel_us_at_end": round(ku, 2), "trigger": w.extra.get("trigger_step_ms")}), flush=True)
