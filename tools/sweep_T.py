#!/usr/bin/env python3
"""Key/value SpMV time vs compressed length T at fixed head count: intercept = launch + per-wave start-up chain."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mustafar_amd import mustafar_package as mp
from tools.microbench import build_cache, timeit

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
Bp, groups, s = 64, 4, 0.7
BH = Bp * groups
for T in [256, 512, 1024, 2048, 4096, 7936, 16128]:
    ncopies = max(2, int(600e6 // (Bp * T * 118)) + 1)
    ncopies = min(ncopies, 24)
    kcs = [build_cache(Bp, T, s, "key", dev, gen) for _ in range(ncopies)]
    vcs = [build_cache(Bp, T, s, "value", dev, gen) for _ in range(ncopies)]
    ws = torch.zeros(1, dtype=torch.float16, device=dev)
    q = torch.randn((BH, 1, 128), device=dev, generator=gen).half()
    p = torch.softmax(torch.randn((BH, 1, T), device=dev, generator=gen), -1).half()
    st = {"i": 0}
    def rk():
        st["i"] += 1
        mp.mustafar_key_formulation(*kcs[st["i"] % ncopies], q, T, 128, BH, groups)
    def rv():
        st["i"] += 1
        mp.mustafar_value_formulation(*vcs[st["i"] % ncopies], p, ws, 128, T, BH, groups)
    print(json.dumps(dict(T=T, key_us=round(timeit(rk, 30) * 1e6, 1), value_us=round(timeit(rv, 30) * 1e6, 1))), flush=True)
    del kcs, vcs
    torch.cuda.empty_cache()
