"""CPU restatement of the reference's DENSE pruned path -- the baseline BASELINE.md names.

TEST INFRASTRUCTURE ONLY (see mustafar_oracle.c): used by tests/, smoke() and bench.py's cpu_baseline leg.

Follows models/llama_mustafar_Kt_Mag_Vt_Mag.py decode branch: the cache stays a dense fp16 tensor whose pruned
entries are zeros; attn = q @ K^T (:873) / sqrt(d), softmax in fp32 cast back to fp16 (:963), out = p @ V (:974).
"""
from __future__ import annotations

import math
import time

import torch


def dense_decode_layer(q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, groups: int) -> torch.Tensor:
    """q [B,Hq,1,D], K/V [B,Hkv,L,D] (pruned-but-dense, fp16) -> [B,Hq,1,D] fp16."""
    B, Hkv, L, D = K.shape
    # GQA without materialising repeat_kv copies: [B,Hkv,G,1,D] x [B,Hkv,1,D,L]
    qg = q.view(B, Hkv, groups, 1, D)
    att = torch.matmul(qg, K.unsqueeze(2).transpose(-1, -2)) / math.sqrt(D)
    p = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
    out = torch.matmul(p, V.unsqueeze(2))
    return out.view(B, Hkv * groups, 1, D)


def time_dense_cpu(batch, Hq, Hkv, L, D, sparsity_keep, layers_sample=1, repeats=3, seed=42, budget_s=25.0):
    """Time `layers_sample` dense layers on the host CPU (all torch threads); returns seconds per layer and dtype used.

    K/V are synthetic N(0,1) with the pruned fraction zeroed at random positions (the dense path's cost does not
    depend on where the zeros are).  fp16 storage as in the reference; matmuls run in the dtype the CPU backend
    executes fastest among {fp16, fp32-upcast}, decided by a short probe, and that choice is reported.
    """
    g = torch.Generator().manual_seed(seed)
    groups = Hq // Hkv
    Ks, Vs = [], []
    for _ in range(layers_sample):
        K = torch.randn(batch, Hkv, L, D, generator=g).half()
        V = torch.randn(batch, Hkv, L, D, generator=g).half()
        K *= (torch.rand(K.shape, generator=g) < sparsity_keep)
        V *= (torch.rand(V.shape, generator=g) < sparsity_keep)
        Ks.append(K)
        Vs.append(V)
    q = torch.randn(batch, Hq, 1, D, generator=g).half()

    def run(cast):
        for K, V in zip(Ks, Vs):
            if cast is None:
                dense_decode_layer(q, K, V, groups)
            else:   # upcast per step: the cache itself stays fp16, as in the reference
                dense_decode_layer(q.to(cast), K.to(cast), V.to(cast), groups)

    probe = {}
    for cast in (None, torch.float32):
        t0 = time.perf_counter()
        run(cast)
        probe[cast] = time.perf_counter() - t0
    cast = min(probe, key=probe.get)
    times = []
    t_start = time.perf_counter()
    for _ in range(repeats):
        t0 = time.perf_counter()
        run(cast)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    times.sort()
    return times[len(times) // 2] / layers_sample, ("fp16" if cast is None else "fp16->fp32 upcast"), len(times)
