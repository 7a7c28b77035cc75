#!/usr/bin/env python3
"""Sweep Split_K of the value SpMV (HBM-cold: alternates between caches larger than the 256 MiB Infinity Cache)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mustafar_amd import mustafar_package as mp
from tools.microbench import CFG, build_cache, timeit

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
for name in sys.argv[1:] or ["c3", "c5"]:   # SWEEP_SPLITS=1,2,4 picks the Split_K values (0 = the library's own choice)
    Hq, Hkv, s, L, batch = CFG[name]
    T = ((L - 32) // 256) * 256
    Bp, BH, groups = batch * Hkv, batch * Hq, Hq // Hkv
    ncopies = max(1, int(600e6 // (Bp * T * 118)) + 1)
    kcs = [build_cache(Bp, T, s, "key", dev, gen) for _ in range(ncopies)]
    vcs = [build_cache(Bp, T, s, "value", dev, gen) for _ in range(ncopies)]
    ws = torch.zeros(1, dtype=torch.float16, device=dev)
    q = torch.randn((BH, 1, 128), device=dev, generator=gen).half()
    p = torch.softmax(torch.randn((BH, 1, T), device=dev, generator=gen), -1).half()
    state = {"i": 0}
    def runk():
        state["i"] += 1
        mp.mustafar_key_formulation(*kcs[state["i"] % ncopies], q, T, 128, BH, groups)
    tk = timeit(runk, 20)
    print(json.dumps(dict(cfg=name, copies=ncopies, key_us=round(tk * 1e6, 1))), flush=True)
    from mustafar_amd import _lib
    picked = _lib.load().mustafar_value_pick_split_k(128, 1, T, BH, groups)
    print(json.dumps(dict(cfg=name, picked_split_k=picked)), flush=True)
    splits = [int(x) for x in os.environ.get("SWEEP_SPLITS", "1,2,4,8,16,31,62,0").split(",")]
    for sk in splits:
        def runv():
            state["i"] += 1
            mp.mustafar_value_formulation(*vcs[state["i"] % ncopies], p, ws, 128, T, BH, groups, split_k=sk)
        tv = timeit(runv, 20 if sk != 1 else 4)
        print(json.dumps(dict(cfg=name, split_k=sk, value_us=round(tv * 1e6, 1))), flush=True)
    del kcs, vcs
    torch.cuda.empty_cache()
