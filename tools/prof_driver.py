#!/usr/bin/env python3
"""Launch the key and value SpMV a few times on HBM-cold caches (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mustafar_amd import mustafar_package as mp
from tools.microbench import CFG, build_cache

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
Hq, Hkv, s, L, batch = CFG[name]
T = ((L - 32) // 256) * 256
Bp, BH, groups = batch * Hkv, batch * Hq, Hq // Hkv
ncopies = max(2, int(600e6 // (Bp * T * 118)) + 1)
kcs = [build_cache(Bp, T, s, "key", dev, gen) for _ in range(ncopies)]
vcs = [build_cache(Bp, T, s, "value", dev, gen) for _ in range(ncopies)]
ws = torch.zeros(1, dtype=torch.float16, device=dev)
q = torch.randn((BH, 1, 128), device=dev, generator=gen).half()
p = torch.softmax(torch.randn((BH, 1, T), device=dev, generator=gen), -1).half()
calib = torch.empty(128 << 20, dtype=torch.float16, device=dev).normal_()   # 256 MiB: FETCH_SIZE / WRITE_SIZE calibration
for i in range(iters):
    calib_out = calib.clone()
    mp.mustafar_key_formulation(*kcs[i % ncopies], q, T, 128, BH, groups)
    mp.mustafar_value_formulation(*vcs[i % ncopies], p, ws, 128, T, BH, groups)
torch.cuda.synchronize()
