#!/usr/bin/env python3
"""Cost of one 256-token cache append (one layer, K and V) at the BASELINE shapes:

  reference layout : compress the 256 new tokens + the hook's tensor-op append (model :339-390: every bitmap, offset and
                     stream of the layer is re-copied)
  arena            : CompressedArena.append (two kernel passes over the new tokens, one B'-element host read)

    python tools/bench_append.py [--cfg c3 c4] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mustafar_amd import compression
from mustafar_amd.cache import CompressedArena
from mustafar_amd.hook import _compress, append_compressed
from tools.microbench import CFG


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", nargs="+", default=["c3", "c4"])
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(3)
    for name in a.cfg:
        Hq, Hkv, s, L, batch = CFG[name]
        T = ((L - 32) // 256) * 256
        Bp = batch * Hkv

        def pruned(t):
            out = []
            for h0 in range(0, Bp, 16):   # bounded temporaries
                x = torch.randn((min(16, Bp - h0), t, 128), device=dev, generator=gen).half()
                out.append(compression.prune_magnitude(x, s))
            return torch.cat(out)

        base = {w: pruned(T) for w in ("key", "value")}
        blocks = [{w: pruned(256) for w in ("key", "value")} for _ in range(a.reps + 1)]
        ref = {w: _compress(base[w], w) for w in base}
        arena = {w: CompressedArena.from_pruned(base[w], w, cap_tokens=T + 256 * (a.reps + 2)) for w in base}
        del base
        res = {}
        for mode in ("reference_layout", "arena"):
            times = []
            tokens = T
            for i, blk in enumerate(blocks):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for w in ("key", "value"):
                    if mode == "arena":
                        arena[w].append(blk[w])
                    else:
                        ref[w] = append_compressed(ref[w], _compress(blk[w], w), Bp, tokens, 256, 128)
                torch.cuda.synchronize()
                if i:   # first one is the warm-up
                    times.append(time.perf_counter() - t0)
                tokens += 256
            res[mode + "_ms"] = round(1e3 * sum(times) / len(times), 3)
        res.update(cfg=name, heads=Bp, T=T, note="one layer, K + V, 256 new tokens, prune excluded")
        print(json.dumps(res), flush=True)
        del ref, arena, blocks
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
