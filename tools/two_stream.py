#!/usr/bin/env python3
"""Experiment: one decode step of the c3 workload (8 sequences) as ONE chain of full-batch launches vs TWO independent
chains of half-batch launches captured as parallel branches of one hipGraph (one fork and one join per step)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mustafar_amd import _lib
from mustafar_amd.hook import MustafarAttention, MustafarConfig

dev = torch.device("cuda:0")
layers, Hq, Hkv, s, L, batch = int(os.environ.get("LAYERS", 32)), 32, 8, 0.7, 8192, 8
lib = _lib.load()


def build(nb, parts):
    cfg = MustafarConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, k_sparsity=s, v_sparsity=s, api="fused", arena=True)
    attn = [MustafarAttention(cfg) for _ in range(parts)]     # scratch buffers are per object: one per chain
    state, qs, ks, vs = [], [], [], []
    for p in range(parts):
        st_, q_, k_, v_ = [], [], [], []
        for _ in range(layers):
            K = torch.randn(nb, Hkv, L, 128, device=dev).half()
            V = torch.randn(nb, Hkv, L, 128, device=dev).half()
            st_.append(attn[p].to_fused(attn[p].build_cache(K, V)))
            q_.append(torch.randn(nb, Hq, 1, 128, device=dev).half())
            k_.append(torch.randn(nb, Hkv, 1, 128, device=dev).half())
            v_.append(torch.randn(nb, Hkv, 1, 128, device=dev).half())
        state.append(st_); qs.append(q_); ks.append(k_); vs.append(v_)
    return attn, state, qs, ks, vs


def run(parts):
    nb = batch // parts
    attn, state, qs, ks, vs = build(nb, parts)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    for p in range(parts):   # warm-up: scratch allocation outside the capture
        w = state[p][0]
        attn[p].decode_fused(qs[p][0], ks[p][0], vs[p][0], (w[0], w[1].clone(), w[2], w[3].clone(), w[4], w[5]))
    torch.cuda.synchronize()
    side = [torch.cuda.Stream(dev) for _ in range(parts - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream(dev)
        for s_ in side:
            s_.wait_stream(main)
        for p in range(parts):
            ctx = torch.cuda.stream(side[p - 1]) if p else torch.cuda.stream(main)
            with ctx:
                for l in range(layers):
                    attn[p].decode_fused(qs[p][l], ks[p][l], vs[p][l], state[p][l], step_counter=counter)
        for s_ in side:
            main.wait_stream(s_)
        _lib.check(lib.mustafar_counter_add(main.cuda_stream, counter.data_ptr(), 1), "counter")
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"chains={parts} batch/chain={nb}: {dt * 1e3:.3f} ms/step, {batch / dt:.1f} tokens/s", flush=True)


for parts in [int(x) for x in os.environ.get("PARTS", "1,2,4").split(",")]:
    run(parts)
    torch.cuda.empty_cache()
