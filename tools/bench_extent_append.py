#!/usr/bin/env python3
"""Host + device cost of one 256-token trigger of ALL layers when the cache grows by extents, c3 geometry (32 layers, 64 kv-heads):
  per layer  (round 3)  32 x append_extent_pair: two allocations, one compression launch, one flag / length read, two table entries
  batched    (round 4)  append_extent_pairs: one pooled allocation (made ahead by prepare_extents: timed apart), every layer's
                        compression launched back to back by ONE library call, ONE host read, one launch per layer for table + slide
Each line: wall ms of the 32-layer trigger (synchronised), and torch's allocator counters around it (segments the driver was asked for)."""
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


def segs():
    st = torch.cuda.memory_stats(dev)
    return st.get("segment.all.allocated", 0), st.get("num_alloc_retries", 0)


# what a slow call coincides with (round 5: the 33-38 ms stall of "layer 14 of the second repetition", rounds 3 and 4): Python's cyclic garbage
# collector (a full collection walks every tracked object of the process) and the allocator's driver segments are watched per call
import gc
gc_log = []
_gc_t = [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t[0] = time.perf_counter()
    else:
        gc_log.append((info["generation"], (time.perf_counter() - _gc_t[0]) * 1e3, info.get("collected", 0)))


gc.callbacks.append(_gc_cb)
for rep in range(3):
    s0 = segs()
    t0 = time.perf_counter()
    slow = (0.0, -1, "")
    for l in range(layers):
        n_gc, sg = len(gc_log), segs()[0]
        tl = time.perf_counter()
        CompressedArena.append_extent_pair(arenas[l][0], arenas[l][1], wk[l], wk[l], kth, kth)
        dt_l = time.perf_counter() - tl
        what = "; ".join("gc generation %d: %.2f ms (%d collected)" % g for g in gc_log[n_gc:]) or "no gc"
        slow = max(slow, (dt_l, l, what + "; driver segments +%d" % (segs()[0] - sg)))
    torch.cuda.synchronize()
    s1 = segs()
    print("            slowest layer of the repetition: layer %d, %.2f ms  [%s]" % (slow[1], slow[0] * 1e3, slow[2]))
    print("per layer : append_extent_pair x32: %.2f ms   (driver segments allocated during it: %d, allocator retries: %d)"
          % ((time.perf_counter() - t0) * 1e3, s1[0] - s0[0], s1[1] - s0[1]))
# what the FIRST batched trigger of a process pays beyond the later ones, taken apart: a one-layer rehearsal first
import cProfile, pstats
wins1 = [(wk[0].clone(), wk[0].clone())]
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
pool1 = CompressedArena.prepare_extents(arenas[:1], kth, kth)
CompressedArena.append_extent_pairs(arenas[:1], wins1, kth, kth, 288, pool1)
torch.cuda.synchronize(); pr.disable()
print("batched   : one-layer rehearsal (first use in the process): %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(14)
for rep in range(4):
    wins = [(w.clone(), w.clone()) for w in wk]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool = CompressedArena.prepare_extents(arenas, kth, kth)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    s0 = segs()
    CompressedArena.append_extent_pairs(arenas, wins, kth, kth, 288, pool)
    torch.cuda.synchronize()
    s1 = segs()
    print("batched   : append_extent_pairs x32: %.2f ms   (prepare_extents ahead: %.2f ms; driver segments allocated during the trigger: %d)"
          % ((time.perf_counter() - t1) * 1e3, (t1 - t0) * 1e3, s1[0] - s0[0]))
