"""scratch: time the c3 256-step leg (one trigger; extents kernel afterwards) and report the kernel durations per instantiation"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mustafar_amd import _lib, mustafar_package as mp
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
lib = _lib.load(); timer = bench.KernelTimer(mp); timer.install()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = bench.Workload(cfg, 32, dev, 0, 1, None, False, timer, lib)
for rep in range(2):
    dt, (ku, vu, n) = w.timed_graph(300, 1)
    print(json.dumps({"cfg": cfg, "tok_s_300_steps_incl_trigger": round(w.batch * 300 / dt, 1), "kernel_us_after": round(ku, 2), "trigger": w.extra.get("trigger_step_ms")}), flush=True)
    w.extra.pop("trigger_step_ms", None)
