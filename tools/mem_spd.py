#!/usr/bin/env python3
"""mem_spd.py -- the reference's latency / peak-memory harness (mem_spd_test.py) on the attention path of this build.

The reference script builds Meta-Llama-3-8B-Instruct from the HF hub, feeds a batch of 32 prompts of ~300 tokens
('apple bear' * 150, mem_spd_test.py:72-74) and times 1 warm-up + 3 `generate(max_new_tokens=600)` calls, printing the
wall ms per generate and `torch.cuda.max_memory_allocated()` in GB (:81-96).  No weights exist offline and the model
classes are out of scope, so this harness runs the SAME SHAPE of work through the part of the model this build
replaces: per layer, prefill attention + cache construction over the prompt (`MustafarAttention.prefill`, model :405-445),
then `output_length` decode steps (`decode`, model :256-400) -- the cache starts at 256 compressed tokens and two 256-token
compression triggers fire on the way to 900 tokens.  q/k/v are synthetic post-RoPE tensors (seed 42, mem_spd_test.py:63);
projections, MLP and sampling are not run, so the milliseconds are those of the attention path alone.

    python tools/mem_spd.py [--api fused|native|reference ...] [--graph] [--batch 32] [--prompt-length 300]
                            [--output-length 600] [--layers 32] [--repeats 3] [--checkpoint PATH]

`--checkpoint PATH` is accepted for the day a local checkpoint directory exists (a hub NAME is never accepted: there is no
network); this harness does not run the projections, so the weights would not change what it measures.
Prints one JSON line per api, then the reference's two print lines for the last api.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

# mem_spd_test.py:7-10, :22
K_SPARSITY = 0.7
V_SPARSITY = 0.7
GROUP_SIZE = 32
BATCH_SIZE = 32
# Meta-Llama-3-8B-Instruct geometry (mem_spd_test.py:17)
LAYERS, Q_HEADS, KV_HEADS, HEAD_DIM = 32, 32, 8, 128


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--api", nargs="+", default=["fused", "reference"], choices=["fused", "native", "reference"])
    ap.add_argument("--graph", action="store_true", help="api=fused: replay each decode step from a captured hipGraph")
    ap.add_argument("--batch", type=int, default=BATCH_SIZE)
    ap.add_argument("--prompt-length", type=int, default=300)    # :72 ('apple bear' * 150 ~ 300 tokens)
    ap.add_argument("--output-length", type=int, default=600)    # :74
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--repeats", type=int, default=3)            # :81-93: 1 warm-up + 3 timed
    ap.add_argument("--checkpoint", default=None, help="local checkpoint DIRECTORY (never a hub name); see the module docstring")
    return ap.parse_args(argv)


class Generator:
    """prefill + decode loop over `layers` attention layers with synthetic post-RoPE q/k/v."""

    def __init__(self, a, api, dev):
        from mustafar_amd.hook import MustafarAttention, MustafarConfig
        self.a, self.api, self.dev = a, api, dev
        self.cfg = MustafarConfig(num_attention_heads=Q_HEADS, num_key_value_heads=KV_HEADS, head_dim=HEAD_DIM, k_sparsity=K_SPARSITY,
                                  v_sparsity=V_SPARSITY, residual_length=GROUP_SIZE, group_size=GROUP_SIZE, api=api, arena=(api == "fused"))
        self.attn = MustafarAttention(self.cfg)
        g = torch.Generator(device=dev).manual_seed(42)       # mem_spd_test.py:63
        B, P = a.batch, a.prompt_length
        rnd = lambda *s: torch.randn(s, device=dev, generator=g).half()
        # one prompt-sized q/k/v set (every layer prefills with it) and one decode token per layer, like a generate() whose
        # weights never change; the values do not matter for the time
        self.pq, self.pk, self.pv = rnd(B, Q_HEADS, P, HEAD_DIM), rnd(B, KV_HEADS, P, HEAD_DIM), rnd(B, KV_HEADS, P, HEAD_DIM)
        self.dq = [rnd(B, Q_HEADS, 1, HEAD_DIM) for _ in range(a.layers)]
        self.dk = [rnd(B, KV_HEADS, 1, HEAD_DIM) for _ in range(a.layers)]
        self.dv = [rnd(B, KV_HEADS, 1, HEAD_DIM) for _ in range(a.layers)]
        self.triggers = 0

    def prefill(self):
        state = []
        for _ in range(self.a.layers):
            _, past = self.attn.prefill(self.pq, self.pk, self.pv)
            state.append(self.attn.to_fused(past) if self.api == "fused" else past)
        return state

    def decode_eager(self, state, steps):
        out = None
        for _ in range(steps):
            for l in range(self.a.layers):
                C = state[l][4]
                out, state[l] = self.attn.decode(self.dq[l], self.dk[l], self.dv[l], state[l])
                self.triggers += (state[l][4] != C) and l == 0
        return out

    def decode_graph(self, state, steps):
        """api=fused: one captured step replayed with a device-side window counter; a trigger step runs eagerly and the
        graph is re-captured (bench.py's timed form)."""
        from mustafar_amd import _lib
        lib = _lib.load()
        attn, L = self.attn, self.a.layers
        counter = torch.zeros(1, dtype=torch.int32, device=self.dev)
        box = {"g": None, "since": 0, "out": None}

        pool = torch.cuda.graph_pool_handle()   # one pool for the graphs of this decode loop (a re-capture reuses the blocks of the graph it
                                                # replaces; the handle dies with the loop: a pool must not outlive its last graph)

        def capture():
            counter.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                for l in range(L):
                    box["out"], _ = attn.decode_fused(self.dq[l], self.dk[l], self.dv[l], state[l], step_counter=counter)
                _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream(self.dev).cuda_stream, counter.data_ptr(), 1), "counter_add")
            box["g"], box["since"] = g, 0

        def until_trigger():
            p = state[0]
            return 256 - ((p[5] + box["since"] - self.cfg.residual_length - p[4]) % 256)

        capture()
        for _ in range(steps):
            if until_trigger() == 1:
                for l in range(L):
                    state[l] = attn.advance(state[l], box["since"])
                self.decode_eager(state, 1)
                capture()
            else:
                box["g"].replay()
                box["since"] += 1
        for l in range(L):
            state[l] = attn.advance(state[l], box["since"])
        return box["out"]

    def generate(self):
        """One `generate(max_new_tokens=output_length)`: prefill, then the decode steps.  Returns (last output, state)."""
        state = self.prefill()
        if self.api == "fused" and self.a.graph:
            out = self.decode_graph(state, self.a.output_length)
        else:
            out = self.decode_eager(state, self.a.output_length)
        return out, state


def run(a, api, dev, keep=None):
    gen = Generator(a, api, dev)
    if keep is not None:
        keep[api] = gen                                # (tests/test_gpu_mem_spd.py rebuilds the dense answer from the generator's tensors)
    torch.cuda.synchronize(dev)
    gen.generate()                                     # warm-up (mem_spd_test.py:81-83)
    torch.cuda.synchronize(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    times, phases = [], []
    state = out = None
    for _ in range(a.repeats):
        gen.triggers = 0
        state = out = None                             # the previous generate's cache is gone, as in generate() (:92)
        torch.cuda.synchronize(dev)
        st = time.time()                               # :87-93
        state = gen.prefill()
        torch.cuda.synchronize(dev)
        t_pre = time.time() - st
        if api == "fused" and a.graph:
            out = gen.decode_graph(state, a.output_length)
        else:
            out = gen.decode_eager(state, a.output_length)
        torch.cuda.synchronize(dev)
        times.append((time.time() - st) * 1e3)
        phases.append(t_pre * 1e3)
    used_mem = torch.cuda.max_memory_allocated(dev)    # :95
    final = state[0]
    res = {"harness": "mem_spd (attention path only, synthetic q/k/v)", "api": api + ("+graph" if (api == "fused" and a.graph) else ""),
           "batch": a.batch, "prompt_length": a.prompt_length, "output_length": a.output_length, "layers": a.layers,
           "k_sparsity": K_SPARSITY, "v_sparsity": V_SPARSITY, "residual_length": GROUP_SIZE,
           "ms_per_generate": [round(t, 2) for t in times], "ms_per_generate_avg": round(sum(times) / len(times), 2),
           "prefill_ms_avg": round(sum(phases) / len(phases), 2),
           "decode_ms_per_step_avg": round((sum(times) - sum(phases)) / len(times) / max(a.output_length, 1), 4),
           "peak_mem_gb": round(used_mem / 1024 ** 3, 3), "triggers_per_generate": gen.triggers,
           "final_compressed_tokens": int(final[4]), "final_kv_seq_len": int(final[5]), "checkpoint": a.checkpoint}
    return res, out


def main(argv=None, keep=None):
    a = parse(argv)
    if a.checkpoint is not None and not os.path.isdir(a.checkpoint):
        raise SystemExit(f"--checkpoint expects a local directory (got {a.checkpoint!r}); hub names are not accepted: there is no network")
    if not torch.cuda.is_available():
        raise SystemExit("mem_spd.py needs a GPU (the product path has no CPU fallback)")
    dev = torch.device("cuda:0")
    results, outs = [], {}
    for api in a.api:
        res, out = run(a, api, dev, keep)
        results.append(res)
        outs[api] = out
        print(json.dumps(res), flush=True)
    last = results[-1]
    print(f"used time: {last['ms_per_generate_avg']} ms")                 # mem_spd_test.py:94
    print(f"peak mem: {last['peak_mem_gb']} GB")                          # :96
    return results, outs


if __name__ == "__main__":
    main()
