"""Round-6 probe: one fused c3 layer, then the e rows / slabs the launch left (score scratch, workspace) saved to an .npz -- two builds compared with cmp_onepass_state.py."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mustafar_amd import _lib, mustafar_package as mp
lib = _lib.load()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
w = bench.Workload("c3", 1, dev, 0, 1, None, False, bench.KernelTimer(mp), lib)
_lib.check(lib.mustafar_set_fma_engine(2), "eng")
fused = w.fused_state()
fused = [(p[0], p[1].clone(), p[2], p[3].clone(), p[4], p[5]) for p in fused]
got = w.one_step(fused)[0]
torch.cuda.synchronize()
(sc, ws), = w.attn._fused_scratch.values()
S, BH = 36, w.BH
ws_f = ws[: (S * BH * 128 + S * BH * 2) * 4].view(torch.float32)
np.savez(sys.argv[1], out=got.float().cpu().numpy(), sc=sc.view(torch.int16).cpu().numpy(), ws_o=ws_f[: S * BH * 128].view(S, BH, 128).cpu().numpy(),
         ws_ml=ws_f[S * BH * 128: S * BH * 130].view(S, BH, 2).cpu().numpy())
print("saved", sys.argv[1], sc.shape, ws.shape)
