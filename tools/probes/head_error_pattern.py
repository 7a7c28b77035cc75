"""Round-6 probe (profiles/r06_probes.txt item 1): per-head error of the fused step against the two reference entry points at c3 and three short caches."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mustafar_amd import _lib, mustafar_package as mp
lib = _lib.load()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
for name, cfg in (("c3", None), ("tiny", ("tiny", 32, 8, 0.7, 256 + 32, 1)), ("t512", ("t512", 32, 8, 0.7, 512 + 32, 1)), ("t1024", ("t1024", 32, 8, 0.7, 1024 + 32, 2))):
    if cfg: bench.CONFIGS[name] = cfg
    w = bench.Workload(name, 1, dev, 0, 1, None, False, bench.KernelTimer(mp), lib)
    _lib.check(lib.mustafar_set_fma_engine(2), "eng")
    fused = w.fused_state()
    fused = [(p[0], p[1].clone(), p[2], p[3].clone(), p[4], p[5]) for p in fused]
    got = w.one_step(fused)[0].float()
    ch = lib.mustafar_last_decode_choice()
    w.cfg.api, w.cfg.arena = "native", False
    want = w.one_step(list(w.pasts))[0].float()
    e = (got - want).abs()[:, :, 0, :]
    scale = float(want.abs().max())
    print(name, "choice", hex(ch), "T", w.T, "scale %.4f" % scale, "max err per head (x1e4):", [round(float(x) * 1e4, 1) for x in e.amax(dim=(0, 2))][:8], "...")
    print("   per batch:", [round(float(x) * 1e4, 1) for x in e.amax(dim=(1, 2))])
    del w
