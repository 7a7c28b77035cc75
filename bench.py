#!/usr/bin/env python3
"""bench.py -- decode-step throughput of the Mustafar sparse-attention path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--layers 32]

One "step" = one decode step of the attention path over all layers (32): per layer the key SpMV over the
compressed cache, the dense window scores, scale + fp32 softmax, the value SpMV, the dense window p.V
(SURVEY 8d).  Inputs are synthetic N(0,1) K/V/q at the model geometry (no weights exist offline), pruned with the
reference rule and compressed by the HIP kernels; everything is resident in HBM before the timed region.

Workloads: c2 Llama-2-7B 70% L=4096 b1 | c3 Llama-3-8B 70% L=8192 b8 (default: the config the
metric is quoted on) | c4 Llama-3-8B 80% L=32768 b4 | c5 Mistral-7B 70% L=16384 b16 | c1 plumbing.

Multi-GPU: the path does not shard one sequence (north_star) -> independent replicas, one process per GPU,
no data-path collective; `value` = units of all ranks / max-over-ranks time ("scaling": "weak").

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {  # name: (label, Hq, Hkv, sparsity, L, batch)
    "c1": ("Llama-2-7B 50% L=1024 b1 (plumbing)", 32, 32, 0.5, 1024, 1),
    "c2": ("Llama-2-7B 70% L=4096 b1", 32, 32, 0.7, 4096, 1),
    "c3": ("Llama-3-8B 70% L=8192 b8", 32, 8, 0.7, 8192, 8),
    "c4": ("Llama-3-8B 80% L=32768 b4", 32, 8, 0.8, 32768, 4),
    "c5": ("Mistral-7B 70% L=16384 b16", 32, 8, 0.7, 16384, 16),
}
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured streaming ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-api", action="store_true", help="skip the extra timings through the two reference entry points")
    ap.add_argument("--no-graph", action="store_true", help="fused api without hipGraph capture of the step")
    ap.add_argument("--api", default="fused", choices=["fused", "native", "reference"], help="call sequence timed for `value`")
    return ap.parse_args()


def cache_bytes(past):
    k_c, k_w, v_c, v_w, _, _ = past
    n = k_w.numel() * 2 + v_w.numel() * 2
    for c in (k_c, v_c):
        if c is not None:
            n += c[0].numel() * 8 + c[1].numel() * 4 + c[2].flat.numel() * 2 + c[3].numel() * 4
    return n


def algorithmic_bytes(past, BH, which):
    """compressed bytes once per kv-head + dense operand in + result out, per launch."""
    c = past[0] if which == "key" else past[2]
    T = past[4]
    return c[0].numel() * 8 + c[1].numel() * 4 + c[2].flat.numel() * 2 + BH * 128 * 2 + BH * T * 2


class KernelTimer:
    """HIP events around every call of the two operators, recorded on torch's current stream -- the stream the
    C ABI launches on."""

    def __init__(self, mp):
        self.mp = mp
        self.orig = (mp.mustafar_key_formulation, mp.mustafar_value_formulation)
        self.events = {"key": [], "value": []}
        self.enabled = False

    def install(self):
        def wrap(fn, name):
            def inner(*a, **kw):
                if not self.enabled:
                    return fn(*a, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **kw)
                e1.record()
                self.events[name].append((e0, e1))
                return out
            return inner
        self.mp.mustafar_key_formulation = wrap(self.orig[0], "key")
        self.mp.mustafar_value_formulation = wrap(self.orig[1], "value")

    def reset(self):
        self.events = {"key": [], "value": []}

    def avg_us(self, name):
        ev = self.events[name]
        return sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev)) * 1e3, len(ev)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # rehearsal on a one-GPU box: MUSTAFAR_BENCH_REHEARSE=1 puts every rank on cuda:0 and lines them up over gloo
    rehearse = os.environ.get("MUSTAFAR_BENCH_REHEARSE") == "1"
    dev = torch.device("cuda", 0 if rehearse else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist   # RCCL; used only for the barrier and the max-over-ranks of the time
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from mustafar_amd import mustafar_package as mp
    from mustafar_amd.hook import MustafarAttention, MustafarConfig

    label, Hq, Hkv, s, L, batch = CONFIGS[a.config]
    D, R = 128, 32
    T = ((L - R) // 256) * 256
    BH = batch * Hq
    torch.manual_seed(42 + rank)                       # seed of mem_spd_test.py:63 (+rank: replicas differ)
    cfg = MustafarConfig(num_attention_heads=Hq, num_key_value_heads=Hkv, k_sparsity=s, v_sparsity=s,
                         residual_length=R, api="native")
    attn = MustafarAttention(cfg)
    timer = KernelTimer(mp)
    timer.install()

    # ---- build the per-layer caches (resident before the timed region) ------------------------------------------
    torch.cuda.reset_peak_memory_stats(dev)
    pasts, qs, ks, vs = [], [], [], []
    for _ in range(a.layers):
        K = torch.randn(batch, Hkv, L, D, device=dev, dtype=torch.float32).half()
        V = torch.randn(batch, Hkv, L, D, device=dev, dtype=torch.float32).half()
        pasts.append(attn.build_cache(K, V))
        del K, V
        qs.append(torch.randn(batch, Hq, 1, D, device=dev).half())
        ks.append(torch.randn(batch, Hkv, 1, D, device=dev).half())
        vs.append(torch.randn(batch, Hkv, 1, D, device=dev).half())
    kv_bytes = sum(cache_bytes(p) for p in pasts)
    dense_bytes = a.layers * 2 * batch * Hkv * L * D * 2
    alg_key = algorithmic_bytes(pasts[0], BH, "key")
    alg_val = algorithmic_bytes(pasts[0], BH, "value")
    alg_key_all = sum(algorithmic_bytes(p, BH, "key") for p in pasts) / a.layers
    alg_val_all = sum(algorithmic_bytes(p, BH, "value") for p in pasts) / a.layers

    def one_step(state):
        for l in range(a.layers):
            _, state[l] = attn.decode(qs[l], ks[l], vs[l], state[l])

    import ctypes
    from mustafar_amd import _lib
    lib = _lib.load()

    from mustafar_amd.replicas import timed_region

    def bracket(run_steps):
        """barrier + synchronize on both sides, max over ranks (the contract's timed region)."""
        return timed_region(run_steps, dist=dist, device=dev, reduce_on_cpu=rehearse)

    def timed(api, steps, warmup):
        """Eager call sequence `api`; per-kernel HIP events are recorded live inside the timed region."""
        cfg.api = api
        cfg.arena = api == "fused"
        # decode() never mutates a reference-layout past in place; the fused api appends to its windows and to its
        # compressed cache in place, so it gets private copies (to_fused re-houses them in appendable buffers)
        state = [attn.to_fused(p) for p in pasts] if api == "fused" else list(pasts)
        for _ in range(warmup):
            one_step(state)
        timer.reset()
        timer.enabled = api != "fused"
        if api == "fused":
            _lib.check(lib.mustafar_profile_begin(steps * a.layers), "mustafar_profile_begin")
        dt = bracket(lambda: [one_step(state) for _ in range(steps)])
        timer.enabled = False
        if api == "fused":
            ku, vu, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
            _lib.check(lib.mustafar_profile_end(ctypes.byref(ku), ctypes.byref(vu), ctypes.byref(n)), "mustafar_profile_end")
            kern = (ku.value, vu.value, n.value)
        else:
            kern = (timer.avg_us("key")[0], timer.avg_us("value")[0], len(timer.events["key"]))
        return dt, kern

    def timed_graph(steps, warmup):
        """The fused call sequence of a whole step (all layers) captured ONCE into a hipGraph and replayed per step;
        a device-side counter grows the windows between replays.  A step that fires the 256-token compression
        trigger (model :324) runs eagerly and the graph is re-captured after it."""
        cfg.api = "fused"
        cfg.arena = True   # compressed cache in appendable storage: a 256-token trigger writes only the new tokens
        state = [attn.to_fused(p) for p in pasts]
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        warm = [(state[0][0], state[0][1].clone(), state[0][2], state[0][3].clone(), state[0][4], state[0][5])]
        attn.decode_fused(qs[0], ks[0], vs[0], warm[0])          # allocates the scratch buffers outside the capture
        torch.cuda.synchronize(dev)
        box = {"g": None, "since": 0}

        def capture():
            counter.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for l in range(a.layers):
                    attn.decode_fused(qs[l], ks[l], vs[l], state[l], step_counter=counter)
                _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream(dev).cuda_stream, counter.data_ptr(), 1), "counter_add")
            box["g"], box["since"] = g, 0

        def until_trigger():
            p = state[0]
            return 256 - ((p[5] + box["since"] - R - p[4]) % 256)

        def step():
            if until_trigger() == 1:
                for l in range(a.layers):
                    state[l] = attn.advance(state[l], box["since"])
                one_step(state)                  # eager: prune + compress + append inside decode_fused
                capture()
            else:
                box["g"].replay()
                box["since"] += 1

        capture()
        for _ in range(warmup):
            step()
        dt = bracket(lambda: [step() for _ in range(steps)])
        for l in range(a.layers):
            state[l] = attn.advance(state[l], box["since"])
        # per-kernel HIP events: same steps again, eagerly, right after the timed replays (events cannot sit between
        # the kernels of a replayed graph)
        _lib.check(lib.mustafar_profile_begin(steps * a.layers), "mustafar_profile_begin")
        for _ in range(steps):
            one_step(state)
        ku, vu, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _lib.check(lib.mustafar_profile_end(ctypes.byref(ku), ctypes.byref(vu), ctypes.byref(n)), "mustafar_profile_end")
        extra["arena_reserved_bytes"] = int(sum(p[0].bytes_reserved() + p[2].bytes_reserved() + p[1].buf.numel() * 2 + p[3].buf.numel() * 2
                                                for p in state))
        return dt, (ku.value, vu.value, n.value)

    extra = {}
    if a.api == "fused" and not a.no_graph:
        dt, (key_us, val_us, n_key) = timed_graph(a.steps, a.warmup)
    else:
        dt, (key_us, val_us, n_key) = timed(a.api, a.steps, a.warmup)
    n_val = n_key
    API_NOTE = {
        "fused": "mustafar_decode_attention (C ABI extension): key SpMV (+ window scores) -> softmax -> value SpMV (+ window p.V partials) -> sum, one call per layer"
                 + ("" if a.no_graph else "; the whole step captured once in a hipGraph and replayed"),
        "native": "the two reference entry points with un-padded (N=1) operands and a flat stream; PyTorch glue between them",
        "reference": "exact reference call sequence: q/p zero-padded to 8 rows, torch.cat of per-head streams per call, 8-row outputs, PyTorch glue",
    }
    others = {}
    if not a.no_reference_api:
        for api in ("fused", "native", "reference"):
            if api == a.api and (api != "fused" or a.no_graph):
                continue
            st_ = max(2, a.steps // 2)
            dt_o, (ku, vu, _) = timed(api, st_, 1)
            others[api] = {"value": round(world * batch * st_ / dt_o, 2), "unit": "tokens/s", "ms_per_step": round(dt_o / st_ * 1e3, 4),
                           "key_call_us": round(ku, 2), "value_call_us": round(vu, 2),
                           "note": (API_NOTE[api] if api != "fused" else API_NOTE[api].split(";")[0] + "; eager (no graph)")}
    engine_extra = None
    if not a.no_reference_api and a.api == "fused" and not a.no_graph and Hq // Hkv >= 4 and (Hq // Hkv) % 4 == 0:
        # opt-in FMA engine (matrix pipe as a 4-wide FMA unit); NOT the headline: the north_star leaves MFMA off
        _lib.check(lib.mustafar_set_fma_engine(1), "set_fma_engine")
        st_ = max(2, a.steps // 2)
        dt_e, (ku, vu, _) = timed_graph(st_, 1)
        _lib.check(lib.mustafar_set_fma_engine(0), "set_fma_engine")
        engine_extra = {"value": round(world * batch * st_ / dt_e, 2), "unit": "tokens/s", "ms_per_step": round(dt_e / st_ * 1e3, 4),
                        "key_kernel_us": round(ku, 2), "value_kernel_us": round(vu, 2),
                        "note": "same fused call sequence with MUSTAFAR_FMA_ENGINE=mfma (v_mfma_f32_4x4x4_16B_f16 as FMA unit); opt-in, off by default"}
    alloc_peak = torch.cuda.max_memory_allocated(dev)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded live in the timed region above) ------------------
    dom = "value" if val_us >= key_us else "key"
    dom_us = max(val_us, key_us)
    dom_bytes = alg_val_all if dom == "value" else alg_key_all
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # PMC-derived bytes per launch, measured offline
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(a.config, {}).get(dom)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": f"{dom}_spmv_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_us": round(dom_us, 2), "launches_timed": n_val if dom == "value" else n_key,
                "other": {"kernel": ("key" if dom == "value" else "value") + "_spmv_kernel",
                          "avg_launch_us": round(min(val_us, key_us), 2),
                          "achieved": round((alg_key_all if dom == "value" else alg_val_all) / (min(val_us, key_us) * 1e-6) / 1e9, 1)},
                "frac_of_measured_stream_ceiling_6290": round(achieved / 6290.0, 4)}

    # ---- host-CPU dense baseline (oracle/dense_ref.py: the reference's dense pruned path), bounded sample --------
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        from oracle.dense_ref import time_dense_cpu
        keep = 1.0 - (max(1, int(s * D)) - 1) / D
        sample_layers = 1
        t_layer, how, reps = time_dense_cpu(batch, Hq, Hkv, L, D, keep, layers_sample=sample_layers, repeats=3)
        cpu = {"value": round(batch / (t_layer * a.layers), 4), "unit": "tokens/s", "cores": torch.get_num_threads(),
               "kind": "port",
               "sample": f"{sample_layers} of {a.layers} layers of the same workload ({label}), dense pruned q.K^T/sqrt(d) -> fp32 softmax -> p.V "
                         f"in PyTorch on the host ({how}), median of {reps} runs after a probe, scaled x{a.layers}/{sample_layers}"}

    out = {
        "metric": "decode_tokens_per_sec", "value": round(world * batch * a.steps / dt, 2), "unit": "tokens/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": label, "id": a.config, "layers": a.layers, "q_heads": Hq, "kv_heads": Hkv, "head_dim": D,
                   "sparsity": s, "seq_len": L, "compressed_tokens": T, "batch_per_gpu": batch, "residual_length": R,
                   "api": a.api, "api_note": API_NOTE[a.api], "fma_engine": "valu (v_fma_mix_f32; MFMA off)",
                   "parallelism": f"replicas x{world}"},
        "peak_kv_bytes": int(kv_bytes), "dense_kv_bytes": int(dense_bytes), "kv_compression_ratio": round(dense_bytes / kv_bytes, 3),
        "allocator_peak_bytes": int(alloc_peak),
        "allocator_note": "peak of the whole bench process: the reference-layout caches kept for the other call sequences + the "
                          "appendable (arena) copy the timed fused leg runs on + transients; arena_reserved_bytes = what the fused leg holds",
        "arena_reserved_bytes": extra.get("arena_reserved_bytes"),
        "roofline": roofline, "cpu_baseline": cpu, "other_call_sequences": others, "fma_engine_mfma": engine_extra,
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
