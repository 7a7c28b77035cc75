#!/usr/bin/env python3
"""Host + device cost of one 256-token trigger when the cache grows by extents (cache.py: append_extent_pair), c3 geometry:
32 layers x (prune + compress 256 window rows of K and V into an extent each, one flag / length read, table entry)."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mustafar_amd.cache import CompressedArena
from mustafar_amd import compression
dev = "cuda:0"
B, H, D = 8, 8, 128
kth = compression.kth_from_sparsity(0.7, D)
layers = 32
arenas = []
K0 = torch.randn(B, H, 1024, D, device=dev).half()
for l in range(layers):
    arenas.append(CompressedArena.from_raw_pair(K0, K0, 1024, kth, kth))
    for a in arenas[-1]: a.ext_table
wk = [torch.randn(B, H, 288, D, device=dev).half() for _ in range(layers)]
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for l in range(layers):
        CompressedArena.append_extent_pair(arenas[l][0], arenas[l][1], wk[l], wk[l], kth, kth)
    torch.cuda.synchronize()
    print("append_extent_pair x32: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for l in range(layers):
    CompressedArena.append_extent_pair(arenas[l][0], arenas[l][1], wk[l], wk[l], kth, kth)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
