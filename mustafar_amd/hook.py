"""Host-side mirror of the attention hook in models/llama_mustafar_kernel.py (LlamaFlashAttention_MUSTAFAR).

The reference file targets transformers 4.43 + flash-attn and cannot be imported here; what this module
reproduces is the part of `forward` (:199-457) that sits between RoPE and o_proj -- the only part that touches
the Mustafar operators -- with the same cache tuple, the same trigger rule and the same call sequence:

    prefill (:405-445)  dense causal attention, T = ((L - R)//256)*256, prune+compress [:T], window = the rest
    decode  (:256-400)  window append, key SpMV, dense window scores, /sqrt(d), fp32 softmax, value SpMV,
                        dense window p.V, and every 256th step prune+compress the oldest 256 window tokens

`past` is the reference's 6-tuple (k_compressed, k_local_window, v_compressed, v_local_window,
compressed_length, kv_seq_len) (:445), k_compressed = [bitmaps, idxs, nzs(list per kv-head), nz_offset].

`api="reference"` issues exactly the reference's calls (query/probabilities zero-padded to 8 rows :273/:313,
`torch.cat` of the per-head streams on every call :274/:314, 8-row outputs sliced to row 0 :275/:315).
`api="native"` calls the same two entry points un-padded (N = 1) and keeps the packed stream of all heads in
one flat tensor beside the list, so nothing is re-copied per step.
`api="fused"` replaces the PyTorch glue between the two SpMVs by `mustafar_decode_attention` (C ABI extension):
the local window lives in a preallocated buffer that is appended in place, and one call per layer launches
key SpMV (+ window scores) -> softmax -> value SpMV (+ window p.V partials) -> sum.  All three produce the same output.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib, compression, mustafar_package
from .cache import CompressedArena


def _operator_module():
    """The module behind `mustafar_package.mustafar_{key,value}_formulation` in the unfused call sequences: the compiled PyTorch
    extension of the reference's name (mustafar_amd/dropin, built by setup.py on the C ABI: what the reference hook imports,
    ~5 us of host time per call) when it has been built, else the ctypes mirror (~12 us).  Same checks, same C-ABI calls."""
    global _OPERATORS
    if _OPERATORS is None:
        import importlib.util, glob, os
        _OPERATORS = mustafar_package
        hits = glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin", "mustafar_package*.so"))
        if hits and os.environ.get("MUSTAFAR_OPERATORS", "compiled") != "ctypes":
            try:
                _lib.load()   # (libmustafar_hip.so first: the extension links against it)
                spec = importlib.util.spec_from_file_location("mustafar_package", hits[0])
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _OPERATORS = mod
            except Exception:   # an extension built for another torch: the ctypes mirror does the same calls
                _OPERATORS = mustafar_package
    return _OPERATORS


_OPERATORS = None


@dataclass
class MustafarConfig:
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 128
    k_sparsity: float = 0.7
    v_sparsity: float = 0.7
    residual_length: int = 32     # mem_spd_test.py:9, :22
    group_size: int = 32          # carried by the reference config, unused on the kernel path
    api: str = "native"           # "reference" | "native" | "fused"
    arena: bool = False           # api="fused": keep the compressed cache in CompressedArena objects (in-place append)
    extents: bool = True          # arena: a 256-token trigger adds an extent instead of re-housing the cache (cache.py), where the
                                  # decode launch can read extents (GQA groups % 4 == 0, one-pass pair form)
    arena_slack: float = 0.03       # an arena that appends IN PLACE is housed at (1 + arena_slack) x the rows / stream bytes it holds
                                    # (cache.py: DEFAULT_SLACK); one that grows by extents is housed exactly
    # api="fused": this instance's FMA engine ("dot2" | "valu" | "mfma"; None = the process default) and launch structure
    # ("one_pass" | "two_launch"; None = by size).  Carried in every call's `flags` (include/mustafar_hip.h): two instances in
    # one process run what each of them asks for.
    engine: Optional[str] = None
    structure: Optional[str] = None


def repeat_kv(hidden_states: torch.Tensor, n_rep: int) -> torch.Tensor:
    """transformers' repeat_kv as used at model :278, :316."""
    batch, num_key_value_heads, slen, head_dim = hidden_states.shape
    if n_rep == 1:
        return hidden_states
    hidden_states = hidden_states[:, :, None, :, :].expand(batch, num_key_value_heads, n_rep, slen, head_dim)
    return hidden_states.reshape(batch, num_key_value_heads * n_rep, slen, head_dim)


class FlatStreams(list):
    """The reference's per-head list of packed-nz tensors.  `streams.flat` is the packed stream of all heads as ONE tensor --
    `torch.cat(streams)`, what the model does on every decode step (:274/:314), which costs nothing when the elements are the
    `StreamPiece`s that `convert_*_batched` returns (compression.py: the pieces of one buffer, in order, ARE that buffer)."""

    def __init__(self, per_head: List[torch.Tensor], flat: Optional[torch.Tensor] = None):
        super().__init__(per_head)
        self._flat = flat

    @property
    def flat(self) -> Optional[torch.Tensor]:
        if self._flat is None and len(self):
            self._flat = torch.cat(self)
        return self._flat


def nz_offset_from_idxs(idxs: torch.Tensor, heads: int) -> torch.Tensor:
    """model :329-331 / :423-425 without the per-head Python loop: nz_offset[i] = sum_{j<i} idxs[j][-1] // 4."""
    last = idxs.view(heads, -1)[:, -1].to(torch.int64) // 4
    off = torch.zeros(heads, dtype=torch.int32, device=idxs.device)
    if heads > 1:
        off[1:] = torch.cumsum(last, 0)[:-1].to(torch.int32)
    return off


def _compress(x: torch.Tensor, which: str):
    """x: pruned [B', t, D] -> [bitmaps, idxs, FlatStreams, nz_offset] (model :328-337, :422-434)."""
    conv = compression.convert_key_batched if which == "key" else compression.convert_value_batched
    bmps, idxs, nzs = conv(x)
    return [bmps, idxs, FlatStreams(nzs), nz_offset_from_idxs(idxs, x.shape[0])]


def append_compressed(old: list, new: list, heads: int, old_tokens: int, new_tokens: int, head_dim: int) -> list:
    """Cache append of model :339-368 (K) / :372-390 (V): shift the new offsets by each head's old total, splice
    bitmaps/offsets per head, concatenate the streams per head, advance nz_offset.  Tensor ops only (the
    reference builds `last_elements` through a Python list of device scalars, :341)."""
    o_bmp, o_idx, o_nz, o_off = old
    n_bmp, n_idx, n_nz, _ = new
    tiles_per_token = head_dim // 64
    n_last = n_idx.view(heads, -1)[:, -1].to(torch.int64) // 4                      # :341-342
    off = o_off.clone()
    if heads > 1:
        off[1:] += torch.cumsum(n_last, 0)[:-1].to(torch.int32)                      # :343-344
    base = o_idx.view(heads, -1)[:, -1:]                                             # :352-353
    idx = torch.cat([o_idx.view(heads, -1)[:, :-1], n_idx.view(heads, -1) + base], dim=1).flatten()   # :356-360
    bmp = torch.cat([o_bmp.view(heads, old_tokens * tiles_per_token),
                     n_bmp.view(heads, new_tokens * tiles_per_token)], dim=1).flatten()               # :364
    per_head = [torch.cat([o_nz[b], n_nz[b]], dim=0) for b in range(heads)]                           # :368
    return [bmp, idx, FlatStreams(per_head), off]


class Window:
    """Dense local window with spare capacity: `buf` [B, Hkv, cap, D], the first `len` rows are valid (api="fused")."""

    def __init__(self, tensor: torch.Tensor, cap: int):
        B, H, w, D = tensor.shape
        self.buf = torch.empty((B, H, max(cap, w), D), dtype=tensor.dtype, device=tensor.device)
        self.buf[:, :, :w] = tensor
        self.len = w

    @property
    def cap(self) -> int:
        return self.buf.shape[2]

    def view(self) -> torch.Tensor:
        return self.buf[:, :, :self.len]

    def reserve(self, n: int):
        if n > self.cap:
            nb = torch.empty((self.buf.shape[0], self.buf.shape[1], n, self.buf.shape[3]), dtype=self.buf.dtype,
                             device=self.buf.device)
            nb[:, :, :self.len] = self.buf[:, :, :self.len]
            self.buf = nb

    def drop_front(self, n: int):
        keep = self.buf[:, :, n:self.len].clone()
        self.len -= n
        self.buf[:, :, :self.len] = keep

    @staticmethod
    def drop_front_pair(k_w: "Window", v_w: "Window", n: int):
        """Slide both windows by n rows with one launch (the rows that stay -- residual_length of them -- move to the front)."""
        keep = k_w.len - n
        if v_w.len != k_w.len or k_w.buf.shape != v_w.buf.shape or not k_w.buf.is_cuda:
            k_w.drop_front(n)
            v_w.drop_front(n)
            return
        L = _lib.load()
        B, H, cap, D = k_w.buf.shape
        with torch.cuda.device(k_w.buf.device):
            err = L.mustafar_window_drop_front(torch.cuda.current_stream(k_w.buf.device).cuda_stream, k_w.buf.data_ptr(), v_w.buf.data_ptr(),
                                               cap * D, B * H, k_w.len, n)
        _lib.check(err, "mustafar_window_drop_front")
        k_w.len = v_w.len = keep

    def clone(self) -> "Window":
        w = Window.__new__(Window)
        w.buf, w.len = self.buf.clone(), self.len
        return w


class MustafarAttention:
    """Prefill + decode attention over the Mustafar cache (no projections, no RoPE: q/k/v arrive post-RoPE)."""

    def __init__(self, config: MustafarConfig):
        self.cfg = config
        self.num_heads = config.num_attention_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.head_dim = config.head_dim
        self.Reduction_Workspace = None   # model :658: a 1-element fp16 tensor shared by all layers

    # ---- pruning (model :77-153) -----------------------------------------------------------------------------
    def dh_prune_key(self, key_states: torch.Tensor, target_sparsity=None) -> torch.Tensor:
        return compression.prune_magnitude(key_states, self.cfg.k_sparsity if target_sparsity is None else target_sparsity)

    def dh_prune_value(self, value_states: torch.Tensor, target_sparsity=None) -> torch.Tensor:
        return compression.prune_magnitude(value_states, self.cfg.v_sparsity if target_sparsity is None else target_sparsity)

    def _slack(self) -> float:
        """Margin an arena is housed with: none where the cache grows by extents (nothing is ever appended in place), the
        configured one where a trigger appends in place (other GQA shapes, two launches asked for)."""
        cfg = self.cfg
        if cfg.extents and _lib.load().mustafar_decode_reads_extents(self.num_key_value_groups, 32,
                                                                     _lib.ENGINE_FLAGS[cfg.engine] | _lib.STRUCTURE_FLAGS[cfg.structure]):
            return 0.0
        return cfg.arena_slack

    def _ws(self, device):
        if self.Reduction_Workspace is None or self.Reduction_Workspace.device != device:
            self.Reduction_Workspace = torch.zeros(1, dtype=torch.float16, device=device)
        return self.Reduction_Workspace

    # ---- prefill (model :405-445) ----------------------------------------------------------------------------
    def build_cache(self, key_states, value_states):
        """The cache-construction half of prefill (model :416-445): k/v [B,Hkv,L,D] fp16 -> past."""
        bsz, _, kv_seq_len, D = key_states.shape
        total_batch_kv = bsz * self.num_key_value_heads
        # :416 computes ((L - R)//256)*256, which is -256 for L < R; clamp at 0.
        compressed_length = max(0, ((kv_seq_len - self.cfg.residual_length) // 256) * 256)
        if compressed_length != 0 and self.cfg.arena and self.cfg.api == "fused":
            # straight from the raw K / V into appendable storage: prune thresholds in registers, no pruned copy (:419-434).
            # In the model K and V arrive as transpose(1, 2) views of [B, L, H, D] projections (model :224-226; RoPE keeps the
            # strides): the kernel reads rows of 128 contiguous halfs per head, so such views are made contiguous first.
            ks = key_states if key_states.is_contiguous() else key_states.contiguous()
            vs = value_states if value_states.is_contiguous() else value_states.contiguous()
            k_compressed, v_compressed = CompressedArena.from_raw_pair(
                ks, vs, compressed_length, compression.kth_from_sparsity(self.cfg.k_sparsity, D),
                compression.kth_from_sparsity(self.cfg.v_sparsity, D), None, self._slack())
            del ks, vs
            k_local_window = key_states[:, :, compressed_length:, :].clone().contiguous()             # :427
            v_local_window = value_states[:, :, compressed_length:, :].clone().contiguous()           # :435
        elif compressed_length != 0:
            k_pruned = self.dh_prune_key(key_states[:, :, :compressed_length, :])                     # :419
            v_pruned = self.dh_prune_value(value_states[:, :, :compressed_length, :])                 # :420
            if self.cfg.arena and self.cfg.api == "fused":   # (unreachable since the branch above takes every arena prefill; kept for callers that set arena late)
                k_compressed = CompressedArena.from_pruned(k_pruned.reshape(total_batch_kv, -1, D), "key", None, self._slack())
                v_compressed = CompressedArena.from_pruned(v_pruned.reshape(total_batch_kv, -1, D), "value", None, self._slack())
            else:
                k_compressed = _compress(k_pruned.reshape(total_batch_kv, -1, D), "key")              # :422-426
                v_compressed = _compress(v_pruned.reshape(total_batch_kv, -1, D), "value")            # :430-434
            k_local_window = key_states[:, :, compressed_length:, :].clone().contiguous()             # :427
            v_local_window = value_states[:, :, compressed_length:, :].clone().contiguous()           # :435
        else:
            k_compressed, k_local_window, v_compressed, v_local_window = None, key_states, None, value_states
        return (k_compressed, k_local_window, v_compressed, v_local_window, compressed_length, kv_seq_len)   # :445

    def prefill(self, query_states, key_states, value_states):
        """q [B,Hq,L,D], k/v [B,Hkv,L,D] fp16 -> (attn_output [B,Hq,L,D], past)."""
        attn_output = F.scaled_dot_product_attention(                                # flash_attn_func, :410-413
            query_states, repeat_kv(key_states, self.num_key_value_groups),
            repeat_kv(value_states, self.num_key_value_groups), is_causal=True)
        return attn_output, self.build_cache(key_states, value_states)

    # ---- fused decode (extension) ---------------------------------------------------------------------------
    def to_fused(self, past):
        """Wrap the two local windows of a reference-layout `past` into appendable buffers."""
        k_c, k_w, v_c, v_w, C, L = past
        if self.cfg.arena and C and not isinstance(k_c, CompressedArena):
            k_c = CompressedArena.from_reference(k_c, "key", C, None, self._slack())
            v_c = CompressedArena.from_reference(v_c, "value", C, None, self._slack())
        if self.cfg.extents and isinstance(k_c, CompressedArena):
            k_c.ext_table, v_c.ext_table   # (exist before any graph that names them is captured: see decode_fused)
        if isinstance(k_w, Window):
            return (k_c, k_w, v_c, v_w, C, L)
        cap = self.cfg.residual_length + 256   # the longest window: the step that fires the trigger (model :324) holds R + 256 rows
        return (k_c, Window(k_w, cap), v_c, Window(v_w, cap), C, L)

    @staticmethod
    def advance(past, n: int):
        """Account on the host for `n` graph replays of a decode_fused(step_counter=...) call."""
        k_c, k_w, v_c, v_w, C, L = past
        k_w.len += n
        v_w.len += n
        return (k_c, k_w, v_c, v_w, C, L + n)

    def _scratch(self, device, BH, ld, ws_bytes):
        """Score scratch and slab workspace of the fused entry point, one pair per (device, stream): calls on different
        streams may overlap, calls on one stream cannot."""
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        pool = self.__dict__.setdefault("_fused_scratch", {})
        sc, ws = pool.get(key, (None, None))
        # Grown geometrically (+ two 256-token triggers of headroom) and never freed while this object lives: a captured graph
        # of an earlier call keeps the old addresses (bench.py records the graph of the NEXT cache length while the current one
        # is still being replayed).  Geometric growth keeps the retired buffers a small multiple of the live one.
        if sc is None or sc.numel() < BH * ld:
            self.__dict__.setdefault("_retired_scratch", []).append(sc)
            sc = torch.empty(max(BH * (ld + 512), (sc.numel() * 5) // 4 if sc is not None else 0), dtype=torch.float16, device=device)
        if ws is None or ws.numel() < ws_bytes:
            self.__dict__.setdefault("_retired_scratch", []).append(ws)
            ws = torch.empty(max(ws_bytes + ws_bytes // 8, (ws.numel() * 5) // 4 if ws is not None else 0, 1 << 20), dtype=torch.uint8, device=device)
        pool[key] = (sc, ws)
        return sc, ws

    def decode_fused(self, query_states, key_states, value_states, past, step_counter: Optional[torch.Tensor] = None,
                     attention_mask: Optional[torch.Tensor] = None, t_device: Optional[torch.Tensor] = None,
                     t_capacity: Optional[int] = None, defer_trigger: bool = False):
        """Same contract as decode() with api="native"; windows are `Window` objects appended in place.

        `attention_mask` is the hook's additive mask [bsz, 1, 1, kv_seq_len] (model :293-301), applied inside the softmax
        kernel exactly as the model does (fp16 add, clamp at finfo.min).  Under graph replay (`step_counter`) the rows must
        be at least `compressed_length + window capacity` long (checked: a shorter mask raises): the kernels read the first
        `kv_seq_len` columns of each row, whatever the step.

        `step_counter` (int32 device tensor, optional) is added to the window length inside the kernels, so that a
        captured graph of this call can be replayed for consecutive steps (advance it with mustafar_counter_add once
        per step); the host-side lengths/`kv_seq_len` of the returned `past` then describe the FIRST replay and the
        256-token trigger is the caller's business (see bench.py).

        `defer_trigger`: a step that reaches the 256-token trigger (model :324) returns with its windows at R + 256 rows and the
        compression NOT run; the caller runs it for all layers at once with `run_triggers()` before the next step (round 4: one
        library call issues every layer's compression, one host read, no allocation when `prepare_triggers()` was called ahead).

        `t_device` (int32 device tensor) + `t_capacity` (graph replay only: `step_counter` is required with them): the compressed
        tokens IN USE as a device quantity and the capacity the launch is sized for (a cache that grows by extents only).  ONE captured graph of the call then serves every compressed
        length up to `t_capacity`: after a trigger (run eagerly, outside the graph) the caller adds 256 to `t_device` and takes
        256 off `step_counter` (tests/test_gpu_extents.py)."""
        cfg = self.cfg
        bsz, _, q_len, D = query_states.shape
        assert q_len == 1 and D == 128
        BH, Bkv, groups = bsz * self.num_heads, bsz * self.num_key_value_heads, self.num_key_value_groups
        k_c, k_w, v_c, v_w, C, _ = self.to_fused(past)
        C_used = C
        if t_device is not None:   # the launch is sized for the capacity; the kernels read the tokens in use from `t_device`
            if step_counter is None:
                raise ValueError("decode_fused: t_device is for captured graphs and needs step_counter (an eager call passes neither)")
            if t_capacity is None or t_capacity < C or (t_capacity - C) % 256 or not isinstance(k_c, CompressedArena) or t_capacity <= k_c.tokens:
                raise ValueError("decode_fused: t_device needs an arena cache and t_capacity = compressed length + a multiple of 256, beyond the base tokens")
            C = t_capacity
        kv_seq_len = past[-1] + 1
        w_len = k_w.len + 1
        k_w.reserve(w_len)
        v_w.reserve(w_len)
        dev = query_states.device
        L = _lib.load()
        split = L.mustafar_value_pick_split_k(128, 1, C, BH, groups) if C else 1
        ld = (C + max(k_w.cap, v_w.cap) + 31) // 32 * 32   # rows on 64-byte lines of their own (one-pass form: mustafar_hip.h)
        scores, ws = self._scratch(dev, BH, ld, L.mustafar_decode_workspace_bytes(C, BH, groups, split))
        out = torch.empty((bsz, self.num_heads, 1, D), dtype=torch.float16, device=dev)
        q = query_states if query_states.is_contiguous() else query_states.contiguous()
        kn = key_states if key_states.is_contiguous() else key_states.contiguous()
        vn = value_states if value_states.is_contiguous() else value_states.contiguous()
        if k_w.cap != v_w.cap:
            raise RuntimeError("key/value windows must have the same capacity")
        mask_ptr, mask_stride = None, 0
        if attention_mask is not None:
            # a replayed graph reads more columns as the window grows: up to the window capacity (the bound ld_scores gets too)
            need = C + w_len if step_counter is None else C + k_w.cap
            if attention_mask.dim() != 4 or attention_mask.shape[:3] != (bsz, 1, q_len) or attention_mask.shape[3] < need or \
                    (step_counter is None and attention_mask.shape[3] != kv_seq_len):
                raise ValueError(f"Attention mask should be of size {(bsz, 1, q_len, kv_seq_len)}, but is {tuple(attention_mask.size())}")   # :294-297
            if attention_mask.dtype != torch.float16 or attention_mask.device != dev:
                raise RuntimeError("attention_mask must be a float16 tensor on the device of the query")
            if attention_mask.stride(3) != 1:
                attention_mask = attention_mask.contiguous()
            mask_ptr, mask_stride = attention_mask.data_ptr(), attention_mask.stride(0) if bsz > 1 else attention_mask.shape[3]
        p = lambda t: t.data_ptr() if t is not None else None
        use_arena = isinstance(k_c, CompressedArena)
        if use_arena:            # a failed asynchronous append is reported before the cache is read again (no host stall)
            k_c.poll()
            v_c.poll()
            if cfg.extents and (k_c._ext_table is None or v_c._ext_table is None):
                # the extent tables must exist BEFORE a graph that names them is captured: created inside a capture they would
                # live in the graph's pool and their zero-fill would be replayed over the entries with every step
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("decode_fused: extent tables missing under graph capture (to_fused() creates them)")
                k_c.ext_table, v_c.ext_table
        tail = (q.data_ptr(), k_w.buf.data_ptr(), v_w.buf.data_ptr(), kn.data_ptr(), vn.data_ptr(), w_len, k_w.cap,
                scores.data_ptr(), ld, out.data_ptr(), ws.data_ptr(), split, C, BH, groups, math.sqrt(D),
                step_counter.data_ptr() if step_counter is not None else None, mask_ptr, mask_stride, self.num_heads,
                _lib.ENGINE_FLAGS[cfg.engine] | _lib.STRUCTURE_FLAGS[cfg.structure])
        flags = tail[-1]
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            if use_arena and C > k_c.tokens:
                # the cache has grown by extents (or, under a capture ahead of a trigger, is about to): base views + device tables
                if not L.mustafar_decode_reads_extents(groups, ld, flags):
                    # an engine / structure switch after the cache grew: this form reads ONE view -- one copy of the cache.  (Not
                    # possible under a capture, nor for a graph captured ahead of a trigger, whose extent does not exist yet.)
                    if torch.cuda.is_current_stream_capturing() or k_c.tokens + 256 * len(k_c.extents) < C or t_device is not None:
                        raise RuntimeError("decode_fused: this launch form cannot read a cache that grows by extents "
                                           "(MustafarConfig(extents=False), or consolidate() the arenas first)")
                    k_c, v_c = k_c.consolidate(), v_c.consolidate()
                    k_c.ext_table, v_c.ext_table
                    err = L.mustafar_decode_attention_view(st, k_c.view_ptr(), v_c.view_ptr(), *tail)
                else:
                    err = L.mustafar_decode_attention_extents(st, k_c.view_ptr(), v_c.view_ptr(), k_c.tokens, k_c.ext_table.data_ptr(),
                                                              v_c.ext_table.data_ptr(), *tail,
                                                              t_device.data_ptr() if t_device is not None else None)
            elif use_arena:
                err = L.mustafar_decode_attention_view(st, k_c.view_ptr(), v_c.view_ptr(), *tail)
            else:
                err = L.mustafar_decode_attention(
                    st, p(k_c[0]) if C else None, p(k_c[2].flat) if C else None, p(k_c[1]) if C else None, p(k_c[3]) if C else None,
                    p(v_c[0]) if C else None, p(v_c[2].flat) if C else None, p(v_c[1]) if C else None, p(v_c[3]) if C else None, *tail)
        _lib.check(err, "mustafar_decode_attention")
        C = C_used   # (the compressed length in use: `C` above was the capacity when t_device sizes the launch)
        if step_counter is not None:
            return out, (k_c, k_w, v_c, v_w, C, kv_seq_len - 1)   # lengths advance with the device counter
        k_w.len = v_w.len = w_len
        if defer_trigger:
            # run_triggers() works on arena caches (a first trigger, C == 0, makes one when cfg.arena says so): anything else would
            # come back with a cache of the wrong kind, or fail inside run_triggers -- refuse here
            if not (use_arena or (cfg.arena and C == 0)):
                raise RuntimeError("decode_fused(defer_trigger=True) needs an arena cache (MustafarConfig(arena=True)); "
                                   "tuple caches run their trigger inside the step")
            return out, (k_c, k_w, v_c, v_w, C, kv_seq_len)
        if (kv_seq_len - cfg.residual_length - C) % 256 == 0 and w_len >= 256:                          # :324
            kth_k = compression.kth_from_sparsity(cfg.k_sparsity, D)
            kth_v = compression.kth_from_sparsity(cfg.v_sparsity, D)
            if use_arena or (cfg.arena and C == 0):
                # prune (:325-326) + compress + append (:328-390) of the raw window rows in one launch
                k_c, k_w, v_c, v_w, C, _ = self._trigger_one(k_c, k_w, v_c, v_w, C, kv_seq_len, kth_k, kth_v)
                return out, (k_c, k_w, v_c, v_w, C, kv_seq_len)
            else:
                k_blk = self.dh_prune_key(k_w.buf[:, :, :256, :]).reshape(Bkv, -1, D)                   # :325
                v_blk = self.dh_prune_value(v_w.buf[:, :, :256, :]).reshape(Bkv, -1, D)                 # :326
                k_new, v_new = _compress(k_blk, "key"), _compress(v_blk, "value")                       # :328-337
                if C == 0:
                    k_c, v_c = k_new, v_new
                else:
                    k_c = append_compressed(k_c, k_new, Bkv, C, 256, D)
                    v_c = append_compressed(v_c, v_new, Bkv, C, 256, D)
                k_w.drop_front(256)                                                                     # :392-393
                v_w.drop_front(256)
            C += 256
        return out, (k_c, k_w, v_c, v_w, C, kv_seq_len)

    # ---- the 256-token trigger of all layers at once (round 4) ----------------------------------------------------------------------
    def trigger_due(self, past) -> bool:
        """The step that produced `past` (its kv_seq_len already counted) reached the trigger of model :324."""
        k_c, k_w, v_c, v_w, C, L = past
        return isinstance(k_w, Window) and (L - self.cfg.residual_length - C) % 256 == 0 and k_w.len >= 256

    def _batched_ok(self, past) -> bool:
        k_c, k_w, v_c, v_w, C, L = past
        flags = _lib.ENGINE_FLAGS[self.cfg.engine] | _lib.STRUCTURE_FLAGS[self.cfg.structure]
        ld = (C + 256 + k_w.cap + 31) // 32 * 32
        return self.cfg.extents and isinstance(k_c, CompressedArena) and C > 0 and k_c.tokens % 256 == 0 and \
            len(k_c.extents) < k_c.MAX_EXTENTS and bool(_lib.load().mustafar_decode_reads_extents(self.num_key_value_groups, ld, flags))

    def prepare_triggers(self, pasts):
        """Allocate and initialise the storage of the coming trigger of every layer ahead of it (one allocation, three launches):
        run_triggers() then allocates nothing.  Returns the pool to hand to run_triggers (None where the batched form does not apply)."""
        if not pasts or not all(self._batched_ok(p) for p in pasts):
            return None
        D = self.head_dim
        return CompressedArena.prepare_extents([(p[0], p[2]) for p in pasts], compression.kth_from_sparsity(self.cfg.k_sparsity, D),
                                               compression.kth_from_sparsity(self.cfg.v_sparsity, D))

    def run_triggers(self, pasts, pool=None):
        """The trigger (model :324-398) of every layer whose last step reached it (decode_fused(defer_trigger=True), or a replayed
        graph of the step accounted with advance()): prune + compress the 256 oldest window rows of K and V into an extent of the
        cache, slide the windows.  All layers in two library calls and one host read (cache.py: append_extent_pairs) where the cache
        grows by extents; layer by layer otherwise.  Returns the new list of pasts."""
        cfg, D = self.cfg, self.head_dim
        kth_k, kth_v = compression.kth_from_sparsity(cfg.k_sparsity, D), compression.kth_from_sparsity(cfg.v_sparsity, D)
        out = list(pasts)
        due = [i for i, p in enumerate(pasts) if self.trigger_due(p)]
        if not due:
            return out
        for i in due:
            if not (isinstance(pasts[i][0], CompressedArena) or (cfg.arena and pasts[i][4] == 0)):
                raise RuntimeError("run_triggers works on arena caches (MustafarConfig(arena=True)); a tuple cache runs its trigger inside decode()")
        batch = [i for i in due if self._batched_ok(pasts[i])]
        wl = {pasts[i][1].len for i in batch}
        if batch and len(wl) == 1 and len({pasts[i][1].buf.shape for i in batch}) == 1:
            CompressedArena.append_extent_pairs([(pasts[i][0], pasts[i][2]) for i in batch], [(pasts[i][1].buf, pasts[i][3].buf) for i in batch],
                                                kth_k, kth_v, wl.pop(), pool if len(batch) == len(pasts) else None)
            for i in batch:
                k_c, k_w, v_c, v_w, C, L = pasts[i]
                k_w.len = v_w.len = k_w.len - 256
                out[i] = (k_c, k_w, v_c, v_w, C + 256, L)
        else:
            batch = []
        for i in due:
            if i in batch:
                continue
            k_c, k_w, v_c, v_w, C, L = pasts[i]
            out[i] = self._trigger_one(k_c, k_w, v_c, v_w, C, L, kth_k, kth_v)
        return out

    def _trigger_one(self, k_c, k_w, v_c, v_w, C, L, kth_k, kth_v):
        """One layer's trigger over an arena cache (the body decode_fused runs when it is not deferred)."""
        cfg = self.cfg
        groups = self.num_key_value_groups
        flags = _lib.ENGINE_FLAGS[cfg.engine] | _lib.STRUCTURE_FLAGS[cfg.structure]
        ld = (C + max(k_w.cap, v_w.cap) + 31) // 32 * 32
        Lb = _lib.load()
        if C == 0:
            k_c, v_c = CompressedArena.from_raw_pair(k_w.buf, v_w.buf, 256, kth_k, kth_v, None, self._slack())
            if cfg.extents:
                k_c.ext_table, v_c.ext_table
        elif cfg.extents and k_c.tokens % 256 == 0 and Lb.mustafar_decode_reads_extents(groups, ld, flags):
            if len(k_c.extents) >= k_c.MAX_EXTENTS:          # table full: one copy of the cache, then extents again
                k_c, v_c = k_c.consolidate(), v_c.consolidate()
                k_c.ext_table, v_c.ext_table
            CompressedArena.append_extent_pair(k_c, v_c, k_w.buf, v_w.buf, kth_k, kth_v)
        else:
            if k_c.extents:
                k_c, v_c = k_c.consolidate(), v_c.consolidate()
            CompressedArena.append_window_pair(k_c, v_c, k_w.buf, v_w.buf, 256, kth_k, kth_v)
        Window.drop_front_pair(k_w, v_w, 256)                                                   # :392-393, in place
        return (k_c, k_w, v_c, v_w, C + 256, L)

    # ---- decode (model :256-400) -----------------------------------------------------------------------------
    def decode(self, query_states, key_states, value_states, past, attention_mask=None):
        """q [B,Hq,1,D], new k/v [B,Hkv,1,D] -> (attn_output [B,Hq,1,D], past)."""
        cfg = self.cfg
        if cfg.api == "fused":
            return self.decode_fused(query_states, key_states, value_states, past, attention_mask=attention_mask)
        if isinstance(past[1], Window):   # a fused cache handed to the unfused path
            past = (past[0], past[1].view(), past[2], past[3].view(), past[4], past[5])
        if isinstance(past[0], CompressedArena):
            past = (past[0].to_reference(), past[1], past[2].to_reference(), past[3], past[4], past[5])
        bsz, _, q_len, D = query_states.shape
        total_batch_size = bsz * self.num_heads
        total_batch_kv = bsz * self.num_key_value_heads
        groups = self.num_key_value_groups
        k_compressed, k_local_window, v_compressed, v_local_window, compressed_length, _ = past
        kv_seq_len = past[-1] + 1                                                                      # :251
        reference_api = cfg.api == "reference"

        k_local_window = torch.cat([k_local_window, key_states], dim=2)                                # :270
        if compressed_length != 0:
            if reference_api:
                padded_query = F.pad(query_states.view(total_batch_size, -1, D), (0, 0, 0, 7), mode="constant", value=0)   # :273
                att_compressed = _operator_module().mustafar_key_formulation(
                    k_compressed[0], torch.cat(k_compressed[2]), k_compressed[1], k_compressed[3], padded_query,
                    compressed_length, D, total_batch_size, groups)                                    # :274
                att_compressed = att_compressed[:, 0:1, :].view(bsz, self.num_heads, 1, compressed_length)   # :275
            else:
                att_compressed = _operator_module().mustafar_key_formulation(
                    k_compressed[0], k_compressed[2].flat, k_compressed[1], k_compressed[3],
                    query_states.reshape(total_batch_size, 1, D), compressed_length, D, total_batch_size, groups
                ).view(bsz, self.num_heads, 1, compressed_length)
            att_local = torch.matmul(query_states, repeat_kv(k_local_window, groups).transpose(2, 3))  # :278
            att_qkfull = torch.cat([att_compressed, att_local], dim=-1)                                # :279
        else:
            att_qkfull = torch.matmul(query_states, repeat_kv(k_local_window, groups).transpose(2, 3))  # :282
        attn_weights = att_qkfull / math.sqrt(D)                                                       # :284
        if attn_weights.size() != (bsz, self.num_heads, q_len, kv_seq_len):
            raise ValueError(f"Attention weights should be of size {(bsz, self.num_heads, q_len, kv_seq_len)}, "
                             f"but is {attn_weights.size()}")                                          # :287-291
        if attention_mask is not None:                                                                 # :293-301
            attn_weights = attn_weights + attention_mask
            attn_weights = torch.max(attn_weights, torch.tensor(torch.finfo(attn_weights.dtype).min))
        attn_weights = F.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query_states.dtype)     # :304

        v_local_window = torch.cat([v_local_window, value_states], dim=2)                              # :309
        if compressed_length != 0:
            if reference_api:
                padded_score = F.pad(attn_weights[:, :, :, :compressed_length].view(total_batch_size, -1, compressed_length),
                                     (0, 0, 0, 7)).contiguous()                                        # :313
                out_c = _operator_module().mustafar_value_formulation(
                    v_compressed[0], torch.cat(v_compressed[2]), v_compressed[1], v_compressed[3], padded_score,
                    self._ws(query_states.device), D, compressed_length, total_batch_size, groups)     # :314
                out_c = out_c[:, 0:1, :].view(bsz, self.num_heads, 1, D)                               # :315
            else:
                score = attn_weights[:, :, :, :compressed_length].reshape(total_batch_size, 1, compressed_length)
                out_c = _operator_module().mustafar_value_formulation(
                    v_compressed[0], v_compressed[2].flat, v_compressed[1], v_compressed[3], score,
                    self._ws(query_states.device), D, compressed_length, total_batch_size, groups
                ).view(bsz, self.num_heads, 1, D)
            out_l = torch.matmul(attn_weights[:, :, :, compressed_length:], repeat_kv(v_local_window, groups))   # :316
            attn_output = out_c + out_l                                                                # :317
        else:
            attn_output = torch.matmul(attn_weights, repeat_kv(v_local_window, groups))                # :320

        # :324 (the reference would hand a <256-token window to the compressor when this fires with a short
        # window, which asserts there; guard instead)
        if (kv_seq_len - cfg.residual_length - compressed_length) % 256 == 0 and k_local_window.shape[2] >= 256:
            k_blk = self.dh_prune_key(k_local_window[:, :, :256, :])                                   # :325
            v_blk = self.dh_prune_value(v_local_window[:, :, :256, :])                                 # :326
            k_new = _compress(k_blk.reshape(total_batch_kv, -1, D), "key")
            v_new = _compress(v_blk.reshape(total_batch_kv, -1, D), "value")
            if compressed_length == 0:                                                                 # :327-337
                k_compressed, v_compressed = k_new, v_new
            else:                                                                                      # :339-390
                k_compressed = append_compressed(k_compressed, k_new, total_batch_kv, compressed_length, 256, D)
                v_compressed = append_compressed(v_compressed, v_new, total_batch_kv, compressed_length, 256, D)
            k_local_window = k_local_window[:, :, 256:, :].clone().contiguous()                        # :392
            v_local_window = v_local_window[:, :, 256:, :].clone().contiguous()                        # :393
            compressed_length = compressed_length + 256                                                # :398

        past = (k_compressed, k_local_window, v_compressed, v_local_window, compressed_length, kv_seq_len)
        return attn_output, past
