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
no data-path collective; `value` = units of all ranks / max-over-ranks time ("scaling": "weak").  Under
torch.distributed.run the ranks come from the environment; a plain `python bench.py --gpus N` starts N fresh
child processes itself (before this process touches the GPU).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects, plus:
  self_check     the timed call sequence against the two reference entry points, every layer, before the timed region
                 (a mismatch ends the run with a non-zero exit code and no line)
  roofline_fma_mix / roofline_mfma   the same measurement on the exact v_fma_mix engine / the opt-in matrix-pipe engine
  configs        c2 / c4 / c5 sub-results (N = 1 only)
  tokens_per_sec_incl_trigger   >= 256 consecutive steps, the 256-token compression trigger included
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import math
import os
import socket
import subprocess
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
    # the metric's own axis (BASELINE.json.metric: Llama-3-8B, 70 %, seq_len 4k-32k; mem_spd_test.py:7-10, :72-74 fix the model and the
    # sparsity and vary the length): c3's geometry and batch at the other three lengths -> `seq_sweep` in the bench line
    "s4": ("Llama-3-8B 70% L=4096 b8", 32, 8, 0.7, 4096, 8),
    "s16": ("Llama-3-8B 70% L=16384 b8", 32, 8, 0.7, 16384, 8),
    "s32": ("Llama-3-8B 70% L=32768 b8", 32, 8, 0.7, 32768, 8),
}
CONFIGS["t8192"] = ("Llama-3-8B 70% T=8192 b8 [tools only]", 32, 8, 0.7, 8192 + 32, 8)
CONFIGS["t8448"] = ("Llama-3-8B 70% T=8448 b8 [tools only]", 32, 8, 0.7, 8448 + 32, 8)
for _t in (8704, 8960, 9216, 9728, 10240, 12288):   # (off the grid of whole rounds of workgroups: tools/quick.py, the `seq_offgrid` leg)
    CONFIGS[f"t{_t}"] = (f"Llama-3-8B 70% T={_t} b8 [tools only]", 32, 8, 0.7, _t + 32, 8)
CONFIGS["b1"] = ("Llama-3-8B 70% L=8192 b1 [tools only]", 32, 8, 0.7, 8192, 1)
CONFIGS["b1s"] = ("Llama-3-8B 70% L=4096 b1 [tools only]", 32, 8, 0.7, 4096, 1)
CONFIGS["b1l"] = ("Llama-3-8B 70% L=32768 b1 [tools only]", 32, 8, 0.7, 32768, 1)
CONFIGS["m8"] = ("Llama-2-7B (MHA) 70% L=8192 b8 [tools only]", 32, 32, 0.7, 8192, 8)
CONFIGS["g2"] = ("GQA-2 (32 q / 16 kv heads) 70% L=8192 b8 [tools only]", 32, 16, 0.7, 8192, 8)
SEQ_SWEEP = ("s4", "c3", "s16", "s32")
OFFGRID = ("t8192", "t8448", "t8704", "t8960")
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured streaming ceiling
D, R = 128, 32

API_NOTE = {
    "fused": "mustafar_decode_attention (C ABI extension), one call per layer; structure chosen by size: one-pass launch (key phase -> softmax step -> "
             "value phase per 64-token block, slabs merged per row) or key SpMV (+ window scores) -> softmax -> value SpMV (+ window p.V partials) -> sum",
    "native": "the two reference entry points (the compiled mustafar_package extension when built, else its ctypes mirror) with un-padded (N=1) operands and a flat stream; PyTorch glue between them",
    "reference": "exact reference call sequence through the compiled mustafar_package extension (the module the reference hook imports; ctypes mirror if it is not built): q/p zero-padded to 8 rows, torch.cat of per-head streams per call, 8-row outputs, PyTorch glue",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-api", action="store_true", help="skip the extra timings through the two reference entry points")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the c2/c4/c5 sub-results")
    ap.add_argument("--no-trigger-leg", action="store_true", help="skip the 256-step leg that contains a compression trigger")
    ap.add_argument("--no-seq-sweep", action="store_true", help="skip the Llama-3-8B 70 % sweep over L = 4k / 16k / 32k (8k is the main result)")
    ap.add_argument("--no-graph", action="store_true", help="fused api without hipGraph capture of the step")
    ap.add_argument("--api", default="fused", choices=["fused", "native", "reference"], help="call sequence timed for `value`")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ multi-GPU entry
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_replicas(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set), one per GPU.  This process has not touched the GPU and never does (torch.cuda.device_count() does not
    initialise it on this image); rank 0's stdout is inherited, so its JSON line is this command's output."""
    rehearse = os.environ.get("MUSTAFAR_BENCH_REHEARSE") == "1" or os.environ.get("MUSTAFAR_BENCH_DRYRUN") == "1"
    if not rehearse:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but {have} GPU(s) are visible (MUSTAFAR_BENCH_REHEARSE=1 stacks the ranks on cuda:0 over gloo)", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:      # a dead rank leaves the others at a barrier: end them (our own children, by handle)
                rc = code
                for q in alive:
                    q.terminate()
    return rc


# ------------------------------------------------------------------------------------------------ helpers
def kernel_source_tag() -> str:
    """Identifies the kernels a PMC traffic figure was measured on (profiles/hbm_traffic.json carries the same tag)."""
    h = hashlib.sha256()
    for f in ("spmv.hip",):   # the decode kernels the traffic is reported for (compress.hip does not launch in the timed region)
        h.update(open(os.path.join(ROOT, "mustafar_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def ref_cache_bytes(past) -> int:
    """Bytes of a cache in the reference layout (SURVEY 8d 'peak KV bytes'): bitmaps + offsets + streams + nz_offset + windows."""
    k_c, k_w, v_c, v_w, _, _ = past
    n = k_w.numel() * 2 + v_w.numel() * 2
    for c in (k_c, v_c):
        if c is not None:
            n += c[0].numel() * 8 + c[1].numel() * 4 + c[2].flat.numel() * 2 + c[3].numel() * 4
    return n


def algorithmic_bytes(past, BH, which) -> int:
    """compressed bytes once per kv-head + dense operand in + result out, per launch."""
    c = past[0] if which == "key" else past[2]
    T = past[4]
    return c[0].numel() * 8 + c[1].numel() * 4 + c[2].flat.numel() * 2 + BH * 128 * 2 + BH * T * 2


class KernelTimer:
    """HIP events around every call of the two operators, recorded on torch's current stream -- the stream the
    C ABI launches on (unfused call sequences only; the fused one has kernel timestamps inside the library)."""

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


class Workload:
    """One BASELINE config resident on the device: per-layer caches (reference layout), fresh q/k/v per layer, and the
    timed call sequences over them."""

    def __init__(self, name, layers, dev, rank, world, dist, rehearse, timer, lib):
        from mustafar_amd.hook import MustafarAttention, MustafarConfig
        self.name, self.layers, self.dev, self.world, self.dist, self.rehearse = name, layers, dev, world, dist, rehearse
        self.timer, self.lib = timer, lib
        self.no_capture_ahead = os.environ.get("MUSTAFAR_BENCH_CAPTURE_AHEAD", "1") == "0"
        self.inflight = int(os.environ.get("MUSTAFAR_BENCH_INFLIGHT", "0"))   # 0: the host queues every replay of the timed region at once
        self.label, self.Hq, self.Hkv, self.s, self.L, self.batch = CONFIGS[name]
        self.T = ((self.L - R) // 256) * 256
        self.BH = self.batch * self.Hq
        torch.manual_seed(42 + rank)                       # seed of mem_spd_test.py:63 (+rank: replicas differ)
        self.cfg = MustafarConfig(num_attention_heads=self.Hq, num_key_value_heads=self.Hkv, k_sparsity=self.s, v_sparsity=self.s,
                                  residual_length=R, api="native")
        self.attn = MustafarAttention(self.cfg)
        self.pasts, self.qs, self.ks, self.vs = [], [], [], []
        for _ in range(layers):
            K = torch.randn(self.batch, self.Hkv, self.L, D, device=dev, dtype=torch.float32).half()
            V = torch.randn(self.batch, self.Hkv, self.L, D, device=dev, dtype=torch.float32).half()
            self.pasts.append(self.attn.build_cache(K, V))
            del K, V
            self.qs.append(torch.randn(self.batch, self.Hq, 1, D, device=dev).half())
            self.ks.append(torch.randn(self.batch, self.Hkv, 1, D, device=dev).half())
            self.vs.append(torch.randn(self.batch, self.Hkv, 1, D, device=dev).half())
        self.ref_kv_bytes = sum(ref_cache_bytes(p) for p in self.pasts)
        self.dense_bytes = layers * 2 * self.batch * self.Hkv * self.L * D * 2
        self.alg_key = sum(algorithmic_bytes(p, self.BH, "key") for p in self.pasts) / layers
        self.alg_val = sum(algorithmic_bytes(p, self.BH, "value") for p in self.pasts) / layers
        self.extra = {}

    # -- call sequences ---------------------------------------------------------------------------------------------
    def one_step(self, state):
        outs = []
        for l in range(self.layers):
            o, state[l] = self.attn.decode(self.qs[l], self.ks[l], self.vs[l], state[l])
            outs.append(o)
        return outs

    def fused_state(self):
        self.cfg.api, self.cfg.arena = "fused", True
        for p in self.pasts:   # every leg starts from the prefilled cache: extents a previous leg's trigger appended are dropped
            for a in (p[0], p[2]):
                if hasattr(a, "drop_extents"):
                    a.drop_extents()
        return [self.attn.to_fused(p) for p in self.pasts]

    def bracket(self, run_steps):
        """barrier + synchronize on both sides, max over ranks (the contract's timed region)."""
        from mustafar_amd.replicas import timed_region
        return timed_region(run_steps, dist=self.dist, device=self.dev, reduce_on_cpu=self.rehearse)

    def profile(self, run, records):
        """Average key / value SpMV kernel duration (the kernels' own start / stop timestamps, what rocprofv3 reports) over `run`."""
        from mustafar_amd import _lib
        _lib.check(self.lib.mustafar_profile_begin(records), "mustafar_profile_begin")
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize(self.dev)
        self.last_profile_wall_s = time.perf_counter() - t0    # wall time of the instrumented pass itself (the kernels' durations belong to THIS execution mode)
        ku, vu, fu, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        _lib.check(self.lib.mustafar_profile_end2(ctypes.byref(ku), ctypes.byref(vu), ctypes.byref(fu), ctypes.byref(n)), "mustafar_profile_end2")
        self.last_finish_us = fu.value                          # the row kernel behind a one-pass launch (0: other structures)
        return ku.value, vu.value, n.value

    def self_check(self):
        """The timed call sequence (fused entry point over the arena cache) against the two reference entry points with
        PyTorch glue, every layer, same inputs; fp16: 2 ulp of the output scale + 1e-4.  Returns the worst excess."""
        fused = self.fused_state()
        fused = [(p[0], p[1].clone(), p[2], p[3].clone(), p[4], p[5]) for p in fused]   # private windows: decode appends in place
        got = self.one_step(fused)
        self.cfg.api, self.cfg.arena = "native", False
        want = self.one_step(list(self.pasts))
        worst = 0.0
        for g, w in zip(got, want):
            scale = max(float(w.float().abs().max()), 2.0 ** -6)
            err = float((g.float() - w.float()).abs().max())
            if not math.isfinite(err):
                return float("inf")
            worst = max(worst, err / (2 * 2.0 ** -11 * scale + 1e-4))
        del fused
        return worst

    def timed_eager(self, api, steps, warmup):
        """Eager call sequence `api`; per-kernel HIP events are recorded live inside the timed region."""
        from mustafar_amd import _lib
        self.cfg.api, self.cfg.arena = api, api == "fused"
        # decode() never mutates a reference-layout past in place; the fused api appends to its windows and to its
        # compressed cache in place, so it gets private copies (to_fused re-houses them in appendable buffers)
        state = self.fused_state() if api == "fused" else list(self.pasts)
        for _ in range(warmup):
            self.one_step(state)
        self.timer.reset()
        self.timer.enabled = False
        dt = self.bracket(lambda: [self.one_step(state) for _ in range(steps)])
        # kernel / call durations: the same steps once more, instrumented, OUTSIDE the timed region -- these call sequences are
        # host-bound, and an event pair around every launch costs them 10-15 % (round 3a timed them instrumented)
        nprof = min(steps, 10)
        if api == "fused":
            return dt, self.profile(lambda: [self.one_step(state) for _ in range(nprof)], nprof * self.layers)
        self.timer.enabled = True
        for _ in range(nprof):
            self.one_step(state)
        torch.cuda.synchronize(self.dev)
        self.timer.enabled = False
        return dt, (self.timer.avg_us("key")[0], self.timer.avg_us("value")[0], len(self.timer.events["key"]))

    def timed_unfused_graph(self, api, steps, warmup):
        """The UNFUSED call sequence `api` ("reference": exactly the hook's -- 8-row pads, torch.cat of the per-head pieces, PyTorch glue,
        the compiled mustafar_package extension; model :270-317) of a whole step, all layers, captured ONCE into a hipGraph and replayed:
        what the drop-in boundary delivers with no Python between the launches.  The sequence reallocates its windows with torch.cat
        every step (model :270, :309), so a captured step has ONE window length: the replays repeat that step (same shapes, same
        inputs), which is what makes the replayed output checkable against the eager one, bit for bit."""
        self.cfg.api, self.cfg.arena = api, False
        state = list(self.pasts)
        want = self.one_step(list(state))                    # eager (also warms allocator pools and the operator's workspace)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            got = self.one_step(list(state))
        for _ in range(warmup):
            g.replay()
        dt = self.bracket(lambda: [g.replay() for _ in range(steps)])
        same = all(torch.equal(a, b) for a, b in zip(got, want))
        return dt, same

    def rehearse_trigger(self):
        """One untimed batched trigger of all layers on throwaway window copies (the extents are dropped again): the first 32-layer
        trigger of a process pays a one-off ~5 ms inside the driver (tools/bench_extent_append.py), which is not a property of the step."""
        from mustafar_amd.cache import CompressedArena
        from mustafar_amd import compression
        state = self.fused_state()
        if not all(isinstance(p[0], CompressedArena) and p[1].len >= 256 for p in state):
            return
        kth_k, kth_v = compression.kth_from_sparsity(self.cfg.k_sparsity, D), compression.kth_from_sparsity(self.cfg.v_sparsity, D)
        pairs = [(p[0], p[2]) for p in state]
        CompressedArena.append_extent_pairs(pairs, [(p[1].buf.clone(), p[3].buf.clone()) for p in state], kth_k, kth_v, state[0][1].len,
                                            CompressedArena.prepare_extents(pairs, kth_k, kth_v))
        torch.cuda.synchronize(self.dev)
        for a, b in pairs:
            a.drop_extents()
            b.drop_extents()

    def timed_graph(self, steps, warmup, start_at_trigger_distance=None, device_t=False):
        """The fused call sequence of a whole step (all layers) captured ONCE into a hipGraph and replayed per step;
        a device-side counter grows the windows between replays.  A step that fires the 256-token compression
        trigger (model :324) runs eagerly; the graph of the steps behind it was captured ahead of it (the cache grows by extents:
        nothing a graph holds moves).  device_t=True: ONE graph for the whole leg -- the launches are sized for a capacity and read
        the compressed tokens in use from a device int (mustafar_decode_attention_extents: T_device); after the trigger step 256
        are added to it and taken off the window counter."""
        from mustafar_amd import _lib
        attn, qs, ks, vs, layers, dev, lib = self.attn, self.qs, self.ks, self.vs, self.layers, self.dev, self.lib
        state = self.fused_state()
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        warm = [(state[0][0], state[0][1].clone(), state[0][2], state[0][3].clone(), state[0][4], state[0][5])]
        attn.decode_fused(qs[0], ks[0], vs[0], warm[0])          # allocates the scratch buffers outside the capture
        torch.cuda.synchronize(dev)
        box = {"g": None, "since": 0, "triggers": 0, "next": None}
        one_graph = device_t and all(hasattr(p[0], "extents") and p[0].tokens % 256 == 0 for p in state)
        t_dev = torch.tensor([state[0][4]], dtype=torch.int32, device=dev) if one_graph else None
        t_cap = state[0][4] + 256 * ((steps + warmup) // 256 + 2) if one_graph else None
        kw_t = {"t_device": t_dev, "t_capacity": t_cap} if one_graph else {}
        if one_graph:   # scratch sized for the capacity, outside the capture
            attn.decode_fused(qs[0], ks[0], vs[0], (state[0][0], state[0][1].clone(), state[0][2], state[0][3].clone(), state[0][4], state[0][5]),
                              step_counter=counter, **kw_t)
            torch.cuda.synchronize(dev)

        def signature(st):   # addresses a captured graph holds: a re-housed arena or window makes the graph stale
            # (a cache that grows by extents keeps its base arrays and its extent table where they are: cache.py)
            return tuple((p[0].signature(), p[2].signature(), p[1].buf.data_ptr(), p[3].buf.data_ptr(), p[1].len, p[3].len, p[4], p[5]) for p in st)

        pool = torch.cuda.graph_pool_handle()   # one memory pool for every graph of this leg: a capture behind a trigger then finds
                                                # the blocks of the graph it replaces instead of asking the driver for new ones (16 ms -> 0.7 ms)

        def record(st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                for l in range(layers):
                    attn.decode_fused(qs[l], ks[l], vs[l], st[l], step_counter=counter, **kw_t)
                _lib.check(lib.mustafar_counter_add(torch.cuda.current_stream(dev).cuda_stream, counter.data_ptr(), 1), "counter_add")
            return g

        def capture():
            counter.zero_()
            box["g"], box["since"], box["next"] = record(state), 0, None

        def until_trigger():
            p = state[0]
            return 256 - ((p[5] + box["since"] - R - p[4]) % 256)

        def copy_window(w):   # (a shallow copy: prepare_triggers only looks at lengths and capacities)
            import copy
            return copy.copy(w)

        def capture_ahead():
            """The graph of the steps BEHIND the coming trigger (256 more compressed tokens, windows back at R rows), recorded while
            the current graph is still being replayed -- the host has nothing else to do between replays -- so the trigger step
            does not pay for a capture.  Adopted only if the caches still live at the addresses recorded."""
            import copy
            fut = []
            for p in state:
                kw, vw = copy.copy(p[1]), copy.copy(p[3])
                kw.len = vw.len = R
                fut.append((p[0], kw, p[2], vw, p[4] + 256, p[5] + box["since"] + until_trigger()))
            box["next"] = (record(fut), signature(fut))

        def step():
            if until_trigger() == 1:
                # the trigger step (model :324-398).  Round 4: its decode comes from the graph like every other step (the windows
                # reach R + 256 rows), then ALL layers' prune + compress + extent append run from two library calls with one host
                # read between them, into storage prepared 8 steps ahead (hook.py: run_triggers / prepare_triggers; no allocation here)
                torch.cuda.synchronize(dev)      # (the syncs bracket the trigger step for `trigger_step_ms`; they cost the leg < 0.1 %)
                t0 = time.perf_counter()
                box["g"].replay()
                box["since"] += 1
                for l in range(layers):
                    state[l] = attn.advance(state[l], box["since"])
                state[:] = attn.run_triggers(state, box.pop("pool", None))
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                box["triggers"] += 1
                ahead = box["next"]
                if one_graph:          # the same graph goes on: 256 more tokens in use, the windows slid by 256
                    t_dev.add_(256)
                    counter.add_(-256)
                    box["since"] = 0
                    box["base"] = box.get("base", 0) + 1
                elif ahead is not None and ahead[1] == signature(state):
                    counter.zero_()
                    box["g"], box["since"], box["next"] = ahead[0], 0, None
                else:
                    capture()
                torch.cuda.synchronize(dev)
                self.extra.setdefault("trigger_step_ms", []).append((round((t1 - t0) * 1e3, 3), round((time.perf_counter() - t1) * 1e3, 3),
                                                                     "same graph (device-side T)" if one_graph else
                                                                     "graph recorded ahead" if ahead is not None and box["g"] is ahead[0] else "re-captured"))
            else:
                if until_trigger() == 8 and box["reaches_trigger"]:   # (a leg that ends in front of the trigger prepares nothing: the work would sit in its timed region)
                    if box["next"] is None and not self.no_capture_ahead and not one_graph:
                        capture_ahead()
                    if "pool" not in box:
                        fut = [attn.advance((p[0], copy_window(p[1]), p[2], copy_window(p[3]), p[4], p[5]), box["since"]) for p in state]
                        box["pool"] = attn.prepare_triggers(fut)
                box["g"].replay()
                box["since"] += 1
                if inflight:   # at most `inflight` replays queued behind the one the GPU works on (MUSTAFAR_BENCH_INFLIGHT; see the note at `inflight`)
                    ev = ring[box["since"] % inflight]
                    if ev[1]:
                        ev[0].synchronize()
                    ev[0].record(torch.cuda.current_stream(dev))
                    ev[1] = True

        inflight = self.inflight
        ring = [[torch.cuda.Event(), False] for _ in range(inflight)]
        capture()
        box["reaches_trigger"] = warmup + steps >= until_trigger()
        for _ in range(warmup):
            step()
        dt = self.bracket(lambda: [step() for _ in range(steps)])
        for l in range(layers):
            state[l] = attn.advance(state[l], box["since"])
        # KV bytes of the container as the timed region leaves it (BEFORE the eager profiling pass below, which may run into the next
        # 256-token trigger and would then report a freshly re-housed cache with near-empty windows)
        self.extra["arena_bytes_reserved"] = int(sum(p[0].bytes_reserved() + p[2].bytes_reserved() + p[1].buf.numel() * 2 + p[3].buf.numel() * 2
                                                     for p in state))
        self.extra["arena_bytes_in_use"] = int(sum(p[0].bytes_in_use() + p[2].bytes_in_use() + p[1].len * p[1].buf.shape[0] * p[1].buf.shape[1] * D * 2
                                                   + p[3].len * p[3].buf.shape[0] * p[3].buf.shape[1] * D * 2 for p in state))
        self.extra["kv_tokens_at_measurement"] = int(state[0][5])
        self.extra["dense_bytes_at_measurement"] = int(layers * 2 * self.batch * self.Hkv * state[0][5] * D * 2)
        # per-kernel durations: the same steps again, eagerly, right after the timed replays (kernel timestamps cannot be
        # taken between the nodes of a replayed graph); rocprofv3 over the replays themselves agrees (profiles/)
        nprof = min(steps, 10)
        kern = self.profile(lambda: [self.one_step(state) for _ in range(nprof)], nprof * layers)
        self.extra["eager_pass_ms_per_step"] = round(self.last_profile_wall_s / nprof * 1e3, 4)
        self.extra["finish_kernel_us"] = round(self.last_finish_us, 2)
        self.extra["triggers_in_timed_region"] = box["triggers"]
        if box["triggers"] and hasattr(state[0][0], "consolidate") and state[0][0].extents:
            # what consolidation costs (a launch form that cannot read extents; a full table: 512 triggers): every layer's base + extents re-housed into one base
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            merged = [(p[0].consolidate(), p[2].consolidate()) for p in state]
            torch.cuda.synchronize(dev)
            self.extra["consolidate_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            del merged
        del state
        return dt, kern

    def fixed_graph(self):
        """The fused step of all layers captured with NO counter step behind it: every replay is the SAME step -- same compressed length, same
        window length, the newest row stored over itself -- for as long as one likes (no trigger is ever reached).  -> (graph, outputs, state)"""
        attn, qs, ks, vs, layers, dev = self.attn, self.qs, self.ks, self.vs, self.layers, self.dev
        state = self.fused_state()
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        warm = (state[0][0], state[0][1].clone(), state[0][2], state[0][3].clone(), state[0][4], state[0][5])
        attn.decode_fused(qs[0], ks[0], vs[0], warm)              # allocates the scratch buffers outside the capture
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = [attn.decode_fused(qs[l], ks[l], vs[l], state[l], step_counter=counter)[0] for l in range(layers)]
        return g, outs, state

    def timed_sustained(self, steps, warmup=5, windows=8):
        """>= 200 replays of the step at FIXED compressed length and window length (no trigger, nothing grows): the rate the device holds once
        the first ~30 ms are over, next to the 20-step headline.  Also the course of it: ms/step over `windows` equal stretches (HIP events
        between replays, on the stream the graph is launched on)."""
        g, _, state = self.fixed_graph()
        dev = self.dev
        for _ in range(warmup):
            g.replay()
        per = max(1, steps // windows)
        evs = []

        def run():
            for i in range(steps):
                if i % per == 0:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(torch.cuda.current_stream(dev))
                    evs.append(e)
                g.replay()
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(dev))
            evs.append(e)
        dt = self.bracket(run)
        marks = [i for i in range(steps) if i % per == 0] + [steps]
        course = [round(evs[i].elapsed_time(evs[i + 1]) / (marks[i + 1] - marks[i]), 4) for i in range(len(evs) - 1)]
        del state
        return dt, course

    def roofline(self, key_us, val_us, n, traffic_file=True):
        """Roofline object of the dominant kernel.  One-pass form (val_us == 0): ONE launch does the key phase, the softmax
        step and the value phase; its algorithmic bytes are those of the two SpMVs together (the e rows it writes
        and reads back stand where the scores out / probabilities in stood)."""
        onepass = val_us == 0
        if onepass:
            # which one-pass kernel ran (the last fused call on this thread was the eager pass that took these timestamps)
            ch = self.lib.mustafar_last_decode_choice()
            eng, form = ch & 15, (ch >> 8) & 15
            dom = {4: f"decode_onepass_small_kernel<{eng}>", 3: f"decode_onepass_sb_kernel<{eng}>", 2: f"decode_onepass_leanpair_kernel<{eng}>", 1: f"decode_onepass_lean_kernel<{eng}>"}.get(
                form, "decode_onepass_kernel<G, matrix pipe>" if eng == 1 else "decode_onepass_kernel<G, v_fma_mix, pair>")
            dom_us, oth_us = key_us, None
            dom_bytes = self.alg_key + self.alg_val
        else:
            dom = "value_spmv_kernel" if val_us >= key_us else "key_spmv_kernel"
            dom_us, oth_us = max(val_us, key_us), min(val_us, key_us)
            dom_bytes = self.alg_val if val_us >= key_us else self.alg_key
            oth_bytes = self.alg_key if val_us >= key_us else self.alg_val
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        traffic, measured_at, items = None, None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # PMC-derived bytes per launch (tools/prof_traffic.sh)
        if traffic_file and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                measured_at = tj.get("kernel_source_tag")
                if measured_at == kernel_source_tag():               # only a figure measured on THESE kernels is reported
                    traffic = tj.get(self.name, {}).get("onepass" if onepass else dom.split("_")[0])
                    items = tj.get(self.name, {}).get("onepass_items") if onepass else None
            except Exception:
                traffic = None
        out = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
               "traffic_note": "PMC bytes per launch, every read priced as a wide read (2 x FETCH_SIZE + WRITE_SIZE): an upper bound -- the 64-byte metadata requests "
                               "are doubled with the rest; traffic_itemised (c3: profiles/r06_traffic_items.txt) prices only the packed streams as wide",
               "traffic_itemised": items,
               "traffic_measured_at": measured_at, "kernel_source_tag": kernel_source_tag(),
               "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_us": round(dom_us, 2), "launches_timed": n,
               "timing_source": "kernel start/stop timestamps (hipExtLaunchKernel events) of an EAGER pass over the same state "
                                "right after the timed graph replays -- kernel timestamps cannot be taken between the nodes of a replayed graph; "
                                "the replays themselves under rocprofv3 --kernel-trace: profiles/ (a few % shorter: no launch gaps)",
               "row_kernel_us": self.extra.get("finish_kernel_us") if onepass else None,
               "eager_pass_ms_per_step": self.extra.get("eager_pass_ms_per_step"),
               "sum_of_kernels_ms_per_step": round(self.layers * (dom_us + (self.extra.get("finish_kernel_us") or 0.0)) * 1e-3, 4) if onepass else None,
               "sum_check": "kernel durations are the kernels' own start/stop timestamps; the intervals of consecutive DEPENDENT kernels overlap by ~1 us each "
                            "(the next kernel's start stamp is taken while the previous one drains), so layers x (one-pass kernel + row kernel) exceeds the "
                            "replayed graph's ms_per_step by 2-3 % and the row kernel's 4 us are mostly the dependent-launch floor, not work; "
                            "eager_pass_ms_per_step is the wall time of the pass the stamps were taken in (host launches between the kernels)",
               "frac_of_measured_stream_ceiling_6290": round(achieved / 6290.0, 4)}
        if not onepass:
            out["other"] = {"kernel": ("key" if dom.startswith("value") else "value") + "_spmv_kernel", "avg_launch_us": round(oth_us, 2),
                            "achieved": round(oth_bytes / (oth_us * 1e-6) / 1e9, 1), "frac": round(oth_bytes / (oth_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
        return out


def prefill_compression_leg(dev, lib, Hkv, s, T, batch):
    """One layer's prefill compression at the workload's shape: prune + bitmaps + offsets + packed streams of K and V in ONE launch
    (mustafar_cache_append_kv -> compress_block_kernel; kernel/compression.py:249-432 + the hook's prune, model :99-110), and the prune
    kernel alone (what an unchanged hook calls first).  Bytes: the raw K and V read once, metadata and streams written once."""
    from mustafar_amd import _lib, compression
    from mustafar_amd.cache import CompressedArena
    Bp = batch * Hkv
    g = torch.Generator(device=dev).manual_seed(7)
    xk = torch.randn((batch, Hkv, T, D), device=dev, generator=g).half()
    xv = torch.randn((batch, Hkv, T, D), device=dev, generator=g).half()
    kth = compression.kth_from_sparsity(s, D)
    ka, va = CompressedArena.from_raw_pair(xk, xv, T, kth, kth)
    scratch = torch.empty(int(lib.mustafar_compress_scratch_bytes(Bp, T)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def fused():
        _lib.check(lib.mustafar_cache_append_kv(st, xk.data_ptr(), xv.data_ptr(), T * D, Bp, T, D, kth, kth, ka.view_ptr(), va.view_ptr(), 0,
                                                ka._totals.data_ptr(), va._totals.data_ptr(), ka.nz_cap, va.nz_cap, ka._overflow.data_ptr(), scratch.data_ptr()), "append_kv")

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n * 1e-3
    t_f = timed(fused)
    assert int(ka._overflow) == 0
    out_k = torch.empty_like(xk)
    t_p = timed(lambda: compression.prune_magnitude(xk.view(Bp, T, D), s, out=out_k.view(Bp, T, D)))
    in_b = (xk.numel() + xv.numel()) * 2
    out_b = int(ka.used.sum() + va.used.sum()) * 2 + 2 * Bp * (2 * T) * 12
    return {"what": "one layer, K and V: prune + compress + pack in one launch (compress_block_kernel), the fused path's prefill; and the prune kernel alone (K)",
            "us": round(t_f * 1e6, 1), "bytes_in": in_b, "bytes_out": out_b, "GBps": round((in_b + out_b) / t_f / 1e9, 1),
            "frac_of_8TBps": round((in_b + out_b) / t_f / (HBM_PEAK_GBPS * 1e9), 3),
            "prune_only_us": round(t_p * 1e6, 1), "prune_only_GBps": round(2 * xk.numel() * 2 / t_p / 1e9, 1),
            "prune_only_frac_of_8TBps": round(2 * xk.numel() * 2 / t_p / (HBM_PEAK_GBPS * 1e9), 3)}


def run_sub_config(name, a, dev, rank, world, dist, rehearse, timer, lib):
    """c2 / c4 / c5 as sub-results of the same line: fused + graph, a few steps, kernel fractions, a self-check."""
    w = Workload(name, a.layers, dev, rank, world, dist, rehearse, timer, lib)
    excess = w.self_check()
    if not excess <= 1.0:
        raise SystemExit(f"bench.py: self-check FAILED at {name}: fused vs reference entry points, {excess:.2f}x the fp16 bound")
    steps = max(3, a.steps // 2)
    dt, (ku, vu, n) = w.timed_graph(steps, 2)
    rl = w.roofline(ku, vu, n)
    out = {"workload": w.label, "value": round(world * w.batch * steps / dt, 2), "unit": "tokens/s", "ms_per_step": round(dt / steps * 1e3, 4),
           "infinity_cache_sized": bool((w.alg_key + w.alg_val) < 256e6 / 4),   # (a layer's bytes against a quarter of the 256 MiB Infinity Cache: cache-resident rather than HBM-bound -- BASELINE.md section 2)
           "steps": steps, "self_check_excess": round(excess, 3),
           "kernel": rl["kernel"], "kernel_us": rl["avg_launch_us"], "roofline_frac": rl["frac"], "roofline_achieved_GBps": rl["achieved"],
           "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"], "traffic": rl["traffic"], "kv_bytes_reference_layout": int(w.ref_kv_bytes), "dense_kv_bytes": int(w.dense_bytes),
           "arena_bytes_in_use": w.extra.get("arena_bytes_in_use"), "arena_bytes_reserved": w.extra.get("arena_bytes_reserved")}
    if (w.Hq // w.Hkv) % 4 != 0:
        # MHA / GQA-2 (c2): >= 256 consecutive steps through the 256-token trigger -- extents and a device-side T exist for every group
        # count since round 4, so the graph recorded ahead (or the SAME graph) goes on behind the trigger here as well
        nst = 256
        w.extra.pop("trigger_step_ms", None)
        w.rehearse_trigger()
        dt_t, _ = w.timed_graph(nst, 1)
        trig = {"value": round(world * w.batch * nst / dt_t, 2), "unit": "tokens/s", "steps": nst, "ms_per_step": round(dt_t / nst * 1e3, 4),
                "triggers": w.extra.get("triggers_in_timed_region"), "trigger_step_ms": w.extra.get("trigger_step_ms")}
        w.extra.pop("trigger_step_ms", None)
        dt_1, _ = w.timed_graph(nst, 1, device_t=True)
        trig["one_graph_device_side_T"] = {"value": round(world * w.batch * nst / dt_1, 2), "ms_per_step": round(dt_1 / nst * 1e3, 4),
                                           "triggers": w.extra.get("triggers_in_timed_region"), "trigger_step_ms": w.extra.get("trigger_step_ms")}
        out["tokens_per_sec_incl_trigger"] = trig
    if (w.Hq // w.Hkv) % 4 == 0:   # the same leg on the other two engines (GQA-4 kernels only), chosen per instance (cfg.engine -> flags)
        for eng, key in (("valu", "fma_engine_fma_mix"), ("mfma", "fma_engine_mfma")):
            w.cfg.engine = eng
            try:
                ex = w.self_check()
                if not ex <= 1.0:
                    raise SystemExit(f"bench.py: self-check FAILED at {name} on engine {eng}: {ex:.2f}x the fp16 bound")
                dt_e, (ku, vu, ne) = w.timed_graph(steps, 2)
                rm = w.roofline(ku, vu, ne, traffic_file=False)
            finally:
                w.cfg.engine = None
            out[key] = {"value": round(world * w.batch * steps / dt_e, 2), "ms_per_step": round(dt_e / steps * 1e3, 4), "self_check_excess": round(ex, 3),
                        "kernel": rm["kernel"], "kernel_us": rm["avg_launch_us"], "roofline_frac": rm["frac"], "roofline_achieved_GBps": rm["achieved"]}
    del w
    torch.cuda.empty_cache()
    return out


def run_seq_point(name, a, dev, rank, world, dist, rehearse, timer, lib):
    """One point of the metric's axis (Llama-3-8B geometry, 70 % / 70 %, batch 8): fused + graph on the default engine, a few steps."""
    w = Workload(name, a.layers, dev, rank, world, dist, rehearse, timer, lib)
    excess = w.self_check()
    if not excess <= 1.0:
        raise SystemExit(f"bench.py: self-check FAILED at {name}: fused vs reference entry points, {excess:.2f}x the fp16 bound")
    steps = max(3, a.steps // 2)
    dt, (ku, vu, n) = w.timed_graph(steps, 2)
    rl = w.roofline(ku, vu, n, traffic_file=False)
    out = seq_point(w.L, w.T, world * w.batch * steps / dt, dt / steps * 1e3, steps, rl, w.extra, excess)
    del w
    torch.cuda.empty_cache()
    return out


def seq_point(L, T, tok_s, ms_step, steps, rl, extra, excess):
    dense = extra.get("dense_bytes_at_measurement")
    return {"seq_len": L, "compressed_tokens": T, "value": round(tok_s, 2), "unit": "tokens/s", "ms_per_step": round(ms_step, 4), "steps": steps,
            "kernel": rl["kernel"], "kernel_us": rl["avg_launch_us"], "roofline_frac": rl["frac"], "roofline_achieved_GBps": rl["achieved"],
            "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"],
            "kv_bytes_in_use": extra.get("arena_bytes_in_use"), "kv_bytes_reserved": extra.get("arena_bytes_reserved"), "dense_kv_bytes": dense,
            "kv_compression_ratio": round(dense / extra["arena_bytes_in_use"], 3) if dense and extra.get("arena_bytes_in_use") else None,
            "self_check_excess": round(excess, 3)}


def main():
    a = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and a.gpus > 1:
        sys.exit(spawn_replicas(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    # rehearsal on a one-GPU box: MUSTAFAR_BENCH_REHEARSE=1 puts every rank on cuda:0 and lines them up over gloo
    rehearse = os.environ.get("MUSTAFAR_BENCH_REHEARSE") == "1"
    if os.environ.get("MUSTAFAR_BENCH_DRYRUN") == "1":     # CPU test of the launch plumbing: ranks, barrier, one line from rank 0
        import torch.distributed as dist
        from mustafar_amd.replicas import timed_region
        if world > 1:
            dist.init_process_group("gloo")
        dt = timed_region(lambda: time.sleep(0.01), dist=dist if world > 1 else None, device=None)
        if rank == 0:
            print(json.dumps({"metric": "decode_tokens_per_sec", "n_gpus": world, "dry_run": True, "seconds": round(dt, 4)}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    dev = torch.device("cuda", 0 if rehearse else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist   # RCCL; used only for the barrier and the max-over-ranks of the time
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from mustafar_amd import _lib, hook as _hook
    lib = _lib.load()
    timer = KernelTimer(_hook._operator_module())   # the module the unfused call sequences go through (compiled extension or ctypes mirror)
    timer.install()

    torch.cuda.reset_peak_memory_stats(dev)
    w = Workload(a.config, a.layers, dev, rank, world, dist, rehearse, timer, lib)

    # ---- self-check of the call sequence about to be timed (untimed) ---------------------------------------------
    excess = w.self_check()
    if not excess <= 1.0:
        raise SystemExit(f"bench.py: self-check FAILED: fused (arena) vs the two reference entry points differ by {excess:.2f}x the fp16 bound")

    # ---- the timed region -----------------------------------------------------------------------------------------
    use_graph = a.api == "fused" and not a.no_graph
    if use_graph:
        dt, (key_us, val_us, n_kern) = w.timed_graph(a.steps, a.warmup)
    else:
        dt, (key_us, val_us, n_kern) = w.timed_eager(a.api, a.steps, a.warmup)
    main_extra = dict(w.extra)

    # ---- the rate the device HOLDS: >= 200 replays of the same step at fixed compressed length and window (no trigger, nothing grows).
    # The headline above is `--steps` replays (20 by default: a ~27-ms burst behind a synchronisation point); a generate is hundreds of steps.
    sustained = None
    if use_graph and not a.no_trigger_leg:
        nsus = 256
        dt_s, course = w.timed_sustained(nsus)
        sustained = {"value": round(world * w.batch * nsus / dt_s, 2), "unit": "tokens/s", "steps": nsus, "ms_per_step": round(dt_s / nsus * 1e3, 4),
                     "vs_headline": round((world * w.batch * nsus / dt_s) / (world * w.batch * a.steps / dt), 4),
                     "ms_per_step_course": course,
                     "note": "256 replays of ONE captured step (T = the config's compressed length, window fixed: no counter step behind the graph), "
                             "timed like the headline; ms_per_step_course = the same run in eight stretches of 32 replays (HIP events on the launch "
                             "stream): the first stretch is the burst the 20-step headline measures, the later ones what the device holds "
                             "(in-kernel clock of both: profiles/r06_clocks.txt)"}

    others = {}
    if not a.no_reference_api:
        for api in ("fused", "native", "reference"):
            if api == a.api and (api != "fused" or a.no_graph):
                continue
            st_ = max(2, a.steps // 2)
            dt_o, (ku, vu, _) = w.timed_eager(api, st_, 1)
            others[api] = {"value": round(world * w.batch * st_ / dt_o, 2), "unit": "tokens/s", "ms_per_step": round(dt_o / st_ * 1e3, 4),
                           "key_call_us": round(ku, 2), "value_call_us": round(vu, 2),
                           "note": API_NOTE[api] + ("; eager (no graph)" if api == "fused" else "")}
        # the reference call sequence once more with the host out of the way: the whole 32-layer step as ONE replayed hipGraph
        try:
            st_ = max(2, a.steps // 2)
            dt_g, same = w.timed_unfused_graph("reference", st_, 2)
            others["reference_graph"] = {"value": round(world * w.batch * st_ / dt_g, 2), "unit": "tokens/s", "ms_per_step": round(dt_g / st_ * 1e3, 4),
                                         "replayed_output_equals_eager": bool(same),
                                         "note": API_NOTE["reference"] + "; the whole step captured once in a hipGraph (torch.cuda.graph around the unchanged "
                                                 "hook code) and replayed: one window length, the same step repeated (INTEGRATION.md)"}
            if not same:
                raise SystemExit("bench.py: the replayed graph of the reference call sequence does not reproduce the eager step")
        except RuntimeError as e:     # (a capture the allocator or an operator refuses is reported, not fatal: the eager legs stand)
            others["reference_graph"] = {"value": None, "error": str(e)[:300]}

    # ---- the other two engines on the same call sequence and timed region, chosen per instance (MustafarConfig.engine -> the call's
    # `flags`): fma_mix = exact fp16 products (the round-1/2 default), mfma = the matrix pipe as a 4-wide FMA unit (opt-in; the
    # north_star leaves MFMA off)
    engine_legs, roofline_legs = {}, {}
    if use_graph and w.Hq // w.Hkv >= 4 and (w.Hq // w.Hkv) % 4 == 0:
        for eng, note in (("valu", "v_fma_mix_f32 per tile and head: exact fp16 x fp16 products, fp16 subnormals included"),
                          ("mfma", "v_mfma_f32_4x4x4_16B_f16 as a 4-wide FMA unit (nothing dense is built); opt-in, off by default")):
            w.cfg.engine = eng
            try:
                ex = w.self_check()
                if not ex <= 1.0:
                    raise SystemExit(f"bench.py: self-check FAILED on engine {eng}: {ex:.2f}x the fp16 bound")
                dt_e, (ku, vu, ne) = w.timed_graph(a.steps, a.warmup)
            finally:
                w.cfg.engine = None
            roofline_legs[eng] = w.roofline(ku, vu, ne, traffic_file=False)
            engine_legs[eng] = {"value": round(world * w.batch * a.steps / dt_e, 2), "unit": "tokens/s", "ms_per_step": round(dt_e / a.steps * 1e3, 4),
                                "steps": a.steps, "key_kernel_us": round(ku, 2), "value_kernel_us": round(vu, 2), "self_check_excess": round(ex, 3),
                                "note": "same fused call sequence, same timed region; " + note}
    engine_extra, roofline_mfma = engine_legs.get("mfma"), roofline_legs.get("mfma")

    # ---- >= 256 consecutive steps: the 256-token compression trigger (prune + compress + in-place append) included --------
    trig = None
    if use_graph and not a.no_trigger_leg:
        nst = 256
        w.extra.pop("trigger_step_ms", None)
        w.rehearse_trigger()
        dt_t, _ = w.timed_graph(nst, 1)
        trig = {"value": round(world * w.batch * nst / dt_t, 2), "unit": "tokens/s", "steps": nst, "ms_per_step": round(dt_t / nst * 1e3, 4),
                "triggers": w.extra.get("triggers_in_timed_region"),
                "trigger_step_ms": w.extra.get("trigger_step_ms"),
                "consolidate_ms": w.extra.get("consolidate_ms"),
                "consolidate_ms_note": "all 32 layers' K and V caches (base + the extent of this leg) re-housed into one base each: what a switch to a launch form that cannot read extents (or a full table: 512 triggers) costs",
                "trigger_step_ms_note": "(the step's decode replayed from the graph + prune/compress of 256 tokens per head into an extent for ALL layers from two library calls and one host read, switch to the graph of the longer cache), wall ms each; a replayed step is ms_per_step",
                "note": "the trigger step: decode from the graph, then every layer's 256 oldest window tokens pruned + compressed into an extent of its cache (storage prepared 8 steps ahead: no allocation in the step); the graph of the steps behind it is recorded 8 steps ahead, between replays"}
        w.extra.pop("trigger_step_ms", None)
        dt_1, _ = w.timed_graph(nst, 1, device_t=True)
        trig["one_graph_device_side_T"] = {"value": round(world * w.batch * nst / dt_1, 2), "unit": "tokens/s", "ms_per_step": round(dt_1 / nst * 1e3, 4),
                                           "triggers": w.extra.get("triggers_in_timed_region"), "trigger_step_ms": w.extra.get("trigger_step_ms"),
                                           "note": "the same leg with ONE captured graph: the launches are sized for a capacity and read the compressed tokens in use from a device int (T_device); no capture behind the first"}
    # ---- a whole generate of the reference harness: 600 new tokens (mem_spd_test.py:72-74 `output_length`) from this prompt length -- three
    # 256-token triggers at L = 8192, and two thirds of the steps at cache lengths off the grid of whole rounds of workgroups -----------------
    gen = None
    if use_graph and not a.no_trigger_leg:
        ngen = 600
        w.extra.pop("trigger_step_ms", None)
        dt_g, _ = w.timed_graph(ngen, 1)
        gen = {"value": round(world * w.batch * ngen / dt_g, 2), "unit": "tokens/s", "steps": ngen, "ms_per_step": round(dt_g / ngen * 1e3, 4),
               "triggers": w.extra.get("triggers_in_timed_region"), "trigger_step_ms": w.extra.get("trigger_step_ms"),
               "vs_headline": round((world * w.batch * ngen / dt_g) / (world * w.batch * a.steps / dt), 4),
               "note": "600 consecutive decode steps from the prompt length of this config (the reference harness's generate: mem_spd_test.py:72-74), "
                       "graph of the longer cache recorded ahead of every trigger; the compressed length grows 7936 -> 8704 on the way"}
    alloc_peak = torch.cuda.max_memory_allocated(dev)

    # ---- the other BASELINE configs as sub-results (N = 1) -----------------------------------------------------------
    label, Hq, Hkv, s, L, batch, T = w.label, w.Hq, w.Hkv, w.s, w.L, w.batch, w.T
    roofline = w.roofline(key_us, val_us, n_kern) if rank == 0 else None
    if roofline is not None and val_us == 0:
        # the same bytes against the STEP: layers x algorithmic bytes of the dominant launch / ms_per_step (row kernel, launch gaps and the dense
        # windows' time included in the denominator, nothing of them in the numerator)
        roofline["step_level_frac"] = round(w.layers * roofline["algorithmic_bytes_per_launch"] / (dt / a.steps) / 1e9 / HBM_PEAK_GBPS, 4)
    ref_kv, dense_bytes = w.ref_kv_bytes, w.dense_bytes
    sub = {}
    if world == 1 and not a.no_other_configs and a.api == "fused" and not a.no_graph:
        del w
        torch.cuda.empty_cache()
        for name in ("c2", "c4", "c5", "b1"):   # (c2's whole layer is under 64 MB: Infinity-Cache-sized, flagged in its entry; b1: the headline's geometry and length at batch 1 -- what single-user decode looks like; not a BASELINE config)
            if name != a.config:
                sub[name] = run_sub_config(name, a, dev, rank, world, dist, rehearse, timer, lib)

    prefill = None
    if world == 1 and not a.no_other_configs and a.api == "fused":
        torch.cuda.empty_cache()
        prefill = prefill_compression_leg(dev, lib, Hkv, s, T, batch)

    # ---- the metric's axis: Llama-3-8B geometry, 70 % / 70 %, batch 8 at L = 4k / 8k / 16k / 32k (N = 1) ------------------------
    sweep = None
    if world == 1 and not a.no_seq_sweep and a.api == "fused" and not a.no_graph and a.config == "c3":
        pts = {}
        for name in SEQ_SWEEP:
            if name == "c3":    # the main result IS the 8k point
                pts["8192"] = seq_point(L, T, world * batch * a.steps / dt, dt / a.steps * 1e3, a.steps, roofline, main_extra, excess)
            else:
                pts[str(CONFIGS[name][4])] = run_seq_point(name, a, dev, rank, world, dist, rehearse, timer, lib)
        sweep = {"workload": "Llama-3-8B geometry (32 q / 8 kv heads, d 128, 32 layers), 70 % K / 70 % V, batch 8, fused entry point + hipGraph, default engine",
                 "points": pts}

    # ---- off the grid of whole rounds of workgroups: c3's geometry at T = 8192 ... 8960 (every BASELINE point sits at T = 2^k - 256, just
    # under one resident round; a cache that has grown by a few 256-token extents does not) -----------------------------------------
    offgrid = None
    if world == 1 and not a.no_seq_sweep and a.api == "fused" and not a.no_graph and a.config == "c3":
        pts = {}
        for name in OFFGRID:
            p_ = run_seq_point(name, a, dev, rank, world, dist, rehearse, timer, lib)
            pts[str(p_["compressed_tokens"])] = {k: p_[k] for k in ("compressed_tokens", "value", "unit", "ms_per_step", "steps", "kernel", "kernel_us", "roofline_frac",
                                                                    "algorithmic_bytes_per_launch", "self_check_excess")}
        offgrid = {"workload": "c3's geometry (Llama-3-8B, 70 % / 70 %, batch 8, 32 layers) at compressed lengths one to four 256-token triggers past T = 7936 "
                               "(2048 + 64 k workgroups of four blocks on the chip's 2048 slots); fused entry point + hipGraph, default engine", "points": pts}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

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

    timed_kv = main_extra.get("arena_bytes_in_use") if use_graph else ref_kv
    out = {
        "metric": "decode_tokens_per_sec", "value": round(world * batch * a.steps / dt, 2), "unit": "tokens/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": label, "id": a.config, "layers": a.layers, "q_heads": Hq, "kv_heads": Hkv, "head_dim": D,
                   "sparsity": s, "seq_len": L, "compressed_tokens": T, "batch_per_gpu": batch, "residual_length": R,
                   "api": a.api, "api_note": API_NOTE[a.api] + ("; the whole step captured once in a hipGraph and replayed" if use_graph else "") + ("; structure run: " + roofline["kernel"] if a.api == "fused" else ""),
                   "fma_engine": "dot2 (v_dot2_f32_f16 on pairs of tiles in the GQA-4 one-pass launch, v_fma_mix_f32 elsewhere; MFMA off)",
                   "parallelism": f"replicas x{world}"},
        "self_check": {"passed": True, "excess_over_fp16_bound": round(excess, 3),
                       "what": "every layer's output of the timed call sequence (fused entry point, arena cache) vs the two reference entry "
                               "points with PyTorch glue on the same inputs; bound = 2 ulp of the output scale + 1e-4"},
        "peak_kv_bytes": int(timed_kv), "peak_kv_bytes_note": "bytes IN USE of the container the timed leg ran on (arena rows + stream regions + windows), taken at the end of the "
                                                              "timed region; kv_bytes_reserved = what is allocated for it (arenas at 1.03 x their content, windows at capacity)",
        "kv_bytes_reserved": main_extra.get("arena_bytes_reserved"), "kv_bytes_reference_layout": int(ref_kv),
        "dense_kv_bytes": int(main_extra.get("dense_bytes_at_measurement") or dense_bytes),
        "kv_tokens_at_measurement": main_extra.get("kv_tokens_at_measurement"),
        "kv_compression_ratio": round((main_extra.get("dense_bytes_at_measurement") or dense_bytes) / timed_kv, 3),
        "kv_compression_ratio_reserved": round((main_extra.get("dense_bytes_at_measurement") or dense_bytes) / main_extra["arena_bytes_reserved"], 3)
        if main_extra.get("arena_bytes_reserved") else None,
        "allocator_peak_bytes": int(alloc_peak),
        "allocator_note": "peak of the whole bench process: the reference-layout caches kept for the other call sequences and the self-check + "
                          "the appendable (arena) copy the timed fused leg runs on + transients",
        "roofline": roofline, "roofline_fma_mix": roofline_legs.get("valu"), "roofline_mfma": roofline_mfma, "cpu_baseline": cpu,
        "other_call_sequences": others, "fma_engine_fma_mix": engine_legs.get("valu"), "fma_engine_mfma": engine_extra, "sustained": sustained, "tokens_per_sec_incl_trigger": trig, "generate_600": gen, "prefill_compression": prefill, "seq_sweep": sweep, "seq_offgrid": offgrid, "configs": sub,
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
