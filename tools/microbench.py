#!/usr/bin/env python3
"""Kernel-level timing of the two SpMV entry points at the BASELINE configs (synthetic N(0,1) K/V).

    python tools/microbench.py [--cfg c2 c3 ...] [--iters 20] [--rows 1 8]

Prints per config: compressed bytes, time per call, achieved algorithmic GB/s (SURVEY 8d formula).
"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from mustafar_amd import compression, mustafar_package as mp

CFG = {  # name: (Hq, Hkv, sparsity, L, batch)
    "c1": (32, 32, 0.5, 1024, 1),
    "c2": (32, 32, 0.7, 4096, 1),
    "c3": (32, 8, 0.7, 8192, 8),
    "c4": (32, 8, 0.8, 32768, 4),
    "c5": (32, 8, 0.7, 16384, 16),
    "b1": (32, 8, 0.7, 8192, 1),      # Llama-3-8B 8k x batch 1 (tools only)
}


def build_cache(Bp, T, s, which, dev, gen, adversarial=False):
    """adversarial: every kept value of a token sits in channels 0..63: V tiles of the upper half and K
    tiles of channels 64..127 are empty, the others carry twice the usual non-zeros."""
    chunk = max(1, (1 << 26) // (T * 128))     # heads per compress call: bounds the dense temporary
    bmps, idxs, nzs = [], [], []
    for b0 in range(0, Bp, chunk):
        b1 = min(Bp, b0 + chunk)
        x = torch.randn((b1 - b0, T, 128), device=dev, generator=gen, dtype=torch.float32).half()
        if adversarial:
            x[:, :, 64:] *= 1e-3     # the magnitude rule then keeps only channels 0..63
        x = compression.prune_magnitude(x, s)
        conv = compression.convert_key_batched if which == "key" else compression.convert_value_batched
        b, i, n = conv(x)
        bmps.append(b); idxs.append(i); nzs.extend(n)
    bmp, idx = torch.cat(bmps), torch.cat(idxs)
    nz = torch.cat(nzs)
    last = (idx[:, -1].long() // 4)
    off = torch.zeros(Bp, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(last, 0)[:-1].int()
    return bmp, nz, idx, off


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", nargs="+", default=["c2", "c3"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rows", nargs="+", type=int, default=[1, 8])
    ap.add_argument("--adversarial", action="store_true", help="all kept values in one 64-channel half (empty + dense tiles)")
    ap.add_argument("--tune", nargs="*", default=[], help="mustafar_tune knobs, knob=value (e.g. 13=0: 8-row value calls on round 1's kernel)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(42)
    if a.tune:
        from mustafar_amd import _lib
        for kv in a.tune:
            k, v = kv.split("=")
            assert _lib.load().mustafar_tune(int(k), int(v)) == 0
    for name in a.cfg:
        Hq, Hkv, s, L, batch = CFG[name]
        T = ((L - 32) // 256) * 256
        Bp, BH, groups = batch * Hkv, batch * Hq, Hq // Hkv
        kc = build_cache(Bp, T, s, "key", dev, gen, a.adversarial)
        vc = build_cache(Bp, T, s, "value", dev, gen, a.adversarial)
        ws = torch.zeros(1, dtype=torch.float16, device=dev)
        meta = Bp * (2 * T * 8 + (2 * T + 1) * 4)
        for N in a.rows:
            q = torch.zeros((BH, N, 128), dtype=torch.float16, device=dev)
            q[:, 0] = torch.randn((BH, 128), device=dev, generator=gen).half()
            p = torch.zeros((BH, N, T), dtype=torch.float16, device=dev)
            p[:, 0] = torch.softmax(torch.randn((BH, T), device=dev, generator=gen), -1).half()
            tk = timeit(lambda: mp.mustafar_key_formulation(*kc, q, T, 128, BH, groups), a.iters)
            tv = timeit(lambda: mp.mustafar_value_formulation(*vc, p, ws, 128, T, BH, groups), a.iters)
            bk = meta + 2 * kc[1].numel() + BH * 128 * 2 + BH * T * 2
            bv = meta + 2 * vc[1].numel() + BH * T * 2 + BH * 128 * 2
            print(json.dumps(dict(cfg=name + ("-adversarial" if a.adversarial else "") + ("".join(" tune " + t for t in a.tune)), rows=N, T=T, Bp=Bp, BH=BH, key_us=round(tk * 1e6, 2), value_us=round(tv * 1e6, 2),
                                  key_alg_MB=round(bk / 1e6, 2), value_alg_MB=round(bv / 1e6, 2),
                                  key_GBps=round(bk / tk / 1e9, 1), value_GBps=round(bv / tv / 1e9, 1))), flush=True)
        del kc, vc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
