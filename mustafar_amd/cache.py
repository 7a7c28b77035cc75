"""Device-resident compressed cache with in-place append.

The reference keeps the compressed cache as `[bitmaps, idxs, nzs(list per kv-head), nz_offset]` and, on every 256-token
trigger, rebuilds all of it: `torch.cat` of the bitmaps and offsets of every head, one `torch.cat` per head for the
streams, a Python list of device scalars for the offsets (models/llama_mustafar_kernel.py:339-390).  `CompressedArena`
holds the same four arrays with spare capacity -- a head's bitmap / offset rows are `cap_tokens` long, every head's
stream has its own region of `nz_cap` halfs -- so an append is two kernel passes over the NEW tokens only
(`mustafar_cache_append_*`) plus one B'-element device->host read that checks the stream regions still fit.

The format is unchanged (bit order, padding to 8, offsets in half2 units, `nz_offset` in uint4 units): the SpMV kernels
read an arena through `mustafar_cache_view` (head strides), and `to_reference()` returns the reference's contiguous list.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch

from . import _lib


# Spare capacity of a new arena: rows for two more 256-token appends, stream regions 3 % over the fullest head's measured
# bytes per token (the heads of one layer differ by < 1 % on i.i.d. data; an append that does not fit re-houses the cache).
# Round 1 reserved t + 1024 tokens x 1.08 = 22 % over the bytes in use at c3; this is ~10 %.
DEFAULT_EXTRA_TOKENS = 512
DEFAULT_HEADROOM = 1.03


class CompressedArena:
    TILES_PER_TOKEN = 2   # head_dim 128 / 64

    def __init__(self, heads: int, which: str, device, cap_tokens: int, nz_cap: int):
        assert which in ("key", "value") and cap_tokens % 64 == 0 and nz_cap % 8 == 0
        self.heads, self.which, self.device = heads, which, device
        self.tokens = 0
        self._alloc(cap_tokens, nz_cap)

    # ---- storage ---------------------------------------------------------------------------------------------
    def _alloc(self, cap_tokens: int, nz_cap: int):
        tiles = cap_tokens * self.TILES_PER_TOKEN
        self.cap_tokens, self.nz_cap = cap_tokens, nz_cap
        self.bmp = torch.empty((self.heads, tiles), dtype=torch.int64, device=self.device)
        self.idx = torch.empty((self.heads, tiles + 1), dtype=torch.int32, device=self.device)
        self.idx[:, 0] = 0
        self.nz = torch.empty((self.heads, nz_cap), dtype=torch.float16, device=self.device)
        self.nz_offset = (torch.arange(self.heads, dtype=torch.int64, device=self.device) * (nz_cap // 8)).to(torch.int32)
        self._totals = torch.empty(self.heads, dtype=torch.int64, device=self.device)
        self._used = torch.zeros(self.heads, dtype=torch.int64)   # host copy of every head's stream length (halfs)
        self._host_totals = torch.zeros(self.heads, dtype=torch.int64).pin_memory() if self.device.type == "cuda" else None
        self._pending = None                                        # event behind an asynchronous copy of _totals into _host_totals
        self._overflow = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._view = _lib.CacheView(self.bmp.data_ptr(), self.nz.data_ptr(), self.idx.data_ptr(), self.nz_offset.data_ptr(),
                                    tiles, tiles + 1,
                                    nz_cap // 8 if (self.heads * (nz_cap // 8) < 2 ** 32 and os.environ.get("MUSTAFAR_NZ_STRIDE", "1") != "0") else 0)

    @property
    def view(self) -> "_lib.CacheView":
        return self._view

    @property
    def used(self) -> torch.Tensor:
        """Exact stream length of every head in halfs (host tensor).  After an asynchronous append (append_window_pair)
        the figures arrive through a pinned-memory copy; reading them here waits for that copy if it is still in flight
        (it was enqueued a trigger period -- 256 decode steps -- ago in the decode loop)."""
        if self._pending is not None:
            self._pending.synchronize()
            self._pending = None
            self._used = self._host_totals.clone()
            if int(self._overflow.item()):
                raise RuntimeError("CompressedArena: a head outgrew its stream region during an asynchronous append "
                                   "(room for one worst-case append is reserved beforehand: this is a bug)")
        return self._used

    @used.setter
    def used(self, value: torch.Tensor):
        self._pending = None
        self._used = value

    def view_ptr(self):
        return ctypes.byref(self._view)

    def bytes_in_use(self) -> int:
        t = self.tokens * self.TILES_PER_TOKEN
        return self.heads * (t * 8 + (t + 1) * 4 + 4) + int(self.used.sum()) * 2

    def bytes_reserved(self) -> int:
        return sum(x.numel() * x.element_size() for x in (self.bmp, self.idx, self.nz, self.nz_offset))

    def _grow(self, cap_tokens: int, nz_cap: int):
        """Re-house the cache with larger rows / regions (amortised: capacities grow geometrically)."""
        self._rehouse(max(cap_tokens, self.cap_tokens), max(nz_cap, self.nz_cap))

    def _rehouse(self, cap_tokens: int, nz_cap: int):
        old = (self.bmp, self.idx, self.nz, self.used.clone(), self.tokens)   # (resolves a pending asynchronous append)
        assert cap_tokens >= self.tokens and nz_cap >= (int(self.used.max()) if self.tokens else 0)
        self._alloc(cap_tokens, nz_cap)
        o_bmp, o_idx, o_nz, used, tokens = old
        t = tokens * self.TILES_PER_TOKEN
        self.bmp[:, :t] = o_bmp[:, :t]
        self.idx[:, :t + 1] = o_idx[:, :t + 1]
        m = int(used.max()) if tokens else 0
        self.nz[:, :m] = o_nz[:, :m]
        self.used, self.tokens = used, tokens

    # ---- append (model :339-390) -----------------------------------------------------------------------------
    def append(self, x: torch.Tensor) -> None:
        """x: pruned fp16 [B', t, 128], t % 64 == 0 -> appended behind the tokens in use."""
        if x.dim() != 3 or x.shape[0] != self.heads or x.shape[2] != 128 or x.shape[1] % 64 or x.dtype != torch.float16:
            raise RuntimeError("CompressedArena.append expects a pruned fp16 [B', t, 128] tensor with t % 64 == 0")
        if not x.is_contiguous():
            x = x.contiguous()
        t = x.shape[1]
        L = _lib.load()
        if self.tokens + t > self.cap_tokens:   # rows full: a quarter more (at least 1024 tokens), the stream regions in proportion
            cap = _round_up(max(self.cap_tokens + max(1024, self.cap_tokens // 4), self.tokens + t), 256)
            self._grow(cap, _round_up(int(self.nz_cap * (cap / float(self.cap_tokens))) + 8, 8))
        st = torch.cuda.current_stream(self.device).cuda_stream
        key = self.which == "key"
        with torch.cuda.device(self.device):
            f = L.mustafar_cache_append_bitmap_key if key else L.mustafar_cache_append_bitmap_value
            _lib.check(f(st, x.data_ptr(), self.heads, t, 128, self.view_ptr(), self.tokens, self._totals.data_ptr()),
                       "mustafar_cache_append_bitmap")
            totals = self._totals.cpu()                  # the one host read of an append (B' values)
            need = int(totals.max())
            if need > self.nz_cap:                       # the new tiles do not fit behind some head's stream: re-house, redo pass 1
                per_token = need / float(self.tokens + t)
                self._grow(self.cap_tokens, _round_up(int(per_token * self.cap_tokens * 1.05) + 1024, 8))
                _lib.check(f(st, x.data_ptr(), self.heads, t, 128, self.view_ptr(), self.tokens, self._totals.data_ptr()),
                           "mustafar_cache_append_bitmap")
            g = L.mustafar_cache_append_pack_key if key else L.mustafar_cache_append_pack_value
            _lib.check(g(st, x.data_ptr(), self.heads, t, 128, self.view_ptr(), self.tokens), "mustafar_cache_append_pack")
        self.used, self.tokens = totals, self.tokens + t

    # ---- fused trigger: prune + compress + append of the RAW window rows of K and V, no host read (model :324-398) ---------
    @staticmethod
    def append_window_pair(k_arena: "CompressedArena", v_arena: "CompressedArena", k_rows: torch.Tensor, v_rows: torch.Tensor,
                           t: int, kth_k: int, kth_v: int) -> None:
        """k_rows / v_rows: fp16 [B, Hkv, >= t, 128] buffers (a window: rows [0, t) of every head are compressed; the row
        stride between heads is the buffer's) holding RAW (unpruned) tokens; kth = max(1, int(sparsity * 128)) (model :97),
        0 for rows that are already pruned.  One launch for both sides (mustafar_cache_append_kv), nothing allocated
        on the device side of the call and nothing read back: room for one worst-case append (t * 128 halfs per head) is
        secured BEFORE the launches from the exact stream lengths of the previous append, which travel to the host through
        an asynchronous pinned-memory copy enqueued right behind it."""
        heads = k_arena.heads
        if v_arena.heads != heads or k_arena.tokens != v_arena.tokens or t % 64 or t <= 0:
            raise RuntimeError("append_window_pair: K and V arenas must describe the same heads and tokens; t % 64 == 0")
        for x in (k_rows, v_rows):
            if x.dtype != torch.float16 or x.dim() != 4 or x.shape[0] * x.shape[1] != heads or x.shape[3] != 128 or x.shape[2] < t \
                    or not x.is_contiguous():
                raise RuntimeError("append_window_pair expects contiguous fp16 [B, Hkv, rows >= t, 128] buffers")
        if k_rows.shape[2] != v_rows.shape[2]:
            raise RuntimeError("append_window_pair: K and V buffers must have the same number of rows per head")
        L = _lib.load()
        for a in (k_arena, v_arena):
            need_rows = a.tokens + t > a.cap_tokens
            need_room = int(a.used.max()) + t * 128 > a.nz_cap          # (a.used waits for the previous append's figures if need be)
            if need_rows or need_room:
                cap = a.cap_tokens
                if need_rows:
                    cap = _round_up(max(cap + max(1024, cap // 4), a.tokens + t), 256)
                per_token = float(a.used.max()) / max(a.tokens, 1) if a.tokens else 72.0
                nz_cap = max(a.nz_cap, _round_up(int(per_token * cap * DEFAULT_HEADROOM) + t * 128 + 1024, 8))
                a._rehouse(cap, nz_cap)
        dev = k_arena.device
        scratch = k_arena._scratch_for(t)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            err = L.mustafar_cache_append_kv(st, k_rows.data_ptr(), v_rows.data_ptr(), k_rows.shape[2] * 128, heads, t, 128, kth_k, kth_v,
                                             k_arena.view_ptr(), v_arena.view_ptr(), k_arena.tokens, k_arena._totals.data_ptr(),
                                             v_arena._totals.data_ptr(), k_arena.nz_cap, v_arena.nz_cap, k_arena._overflow.data_ptr(),
                                             scratch.data_ptr())
            _lib.check(err, "mustafar_cache_append_kv")
            for a in (k_arena, v_arena):
                a._host_totals.copy_(a._totals, non_blocking=True)
                a._pending = torch.cuda.Event()
                a._pending.record(torch.cuda.current_stream(dev))
                a.tokens += t
        v_arena._overflow = k_arena._overflow   # one flag per pair and call (either side's `used` reports it)

    def _scratch_for(self, t: int) -> torch.Tensor:
        n = int(_lib.load().mustafar_compress_scratch_bytes(self.heads, t))
        sc = getattr(self, "_blk_scratch", None)
        if sc is None or sc.numel() < n:
            self._blk_scratch = sc = torch.empty(n, dtype=torch.uint8, device=self.device)
        return sc

    @classmethod
    def from_raw_pair(cls, k_rows: torch.Tensor, v_rows: torch.Tensor, t: int, kth_k: int, kth_v: int, cap_tokens: Optional[int] = None,
                      headroom: float = DEFAULT_HEADROOM):
        """Prefill (model :416-437): prune + compress the first t tokens of raw K / V [B, Hkv, L, 128] into two new arenas,
        one read of the dense block per side and no pruned copy.  The stream regions start from an estimate (kept values
        per token + padding, + 12 %) and are re-housed at the measured size."""
        heads = k_rows.shape[0] * k_rows.shape[1]
        cap = _round_up(cap_tokens if cap_tokens else t + DEFAULT_EXTRA_TOKENS, 256)

        def region(kth: int, worst: bool) -> int:   # halfs per head
            kept = 128 if (worst or kth == 0) else 128 - kth + 1       # without ties; ties keep more (model :107)
            return _round_up(int(t * (kept + 9) * (1.0 if worst else 1.12)) + 2048, 8)

        for worst in (False, True):
            k = cls(heads, "key", k_rows.device, _round_up(t, 64), region(kth_k, worst))
            v = cls(heads, "value", v_rows.device, _round_up(t, 64), region(kth_v, worst))
            k._used = torch.full((heads,), -t * 128, dtype=torch.int64)   # (no room check: an overflow is handled right here)
            v._used = k._used.clone()
            cls.append_window_pair(k, v, k_rows, v_rows, t, kth_k, kth_v)
            try:
                k.used, v.used                       # wait for the stream lengths; raises if a head outgrew its region
                break
            except RuntimeError:
                if worst:
                    raise
        for a in (k, v):
            per_token = float(a.used.max()) / max(t, 1)
            a._rehouse(cap, _round_up(int(per_token * cap * headroom) + 1024, 8))
        return k, v

    # ---- conversion ------------------------------------------------------------------------------------------
    @classmethod
    def from_pruned(cls, x: torch.Tensor, which: str, cap_tokens: Optional[int] = None, headroom: float = DEFAULT_HEADROOM) -> "CompressedArena":
        """Compress x [B', t, 128] (already pruned) into a new arena sized for `cap_tokens` (default: t + DEFAULT_EXTRA_TOKENS)
        with stream regions of `headroom` x the measured halfs per token."""
        heads, t, _ = x.shape
        cap = _round_up(cap_tokens if cap_tokens else t + DEFAULT_EXTRA_TOKENS, 256)
        # first pass into rows of exactly t tokens and a generous guess for the streams (dense would be 128 halfs per
        # token), then re-house at the measured size: the transient is freed, the resident footprint is tight
        a = cls(heads, which, x.device, _round_up(t, 64), _round_up(t * 72 + 1024, 8))
        a.append(x)
        per_token = float(a.used.max()) / max(t, 1)
        nz_cap = _round_up(int(per_token * cap * headroom) + 1024, 8)
        if cap != a.cap_tokens or nz_cap != a.nz_cap:
            a._rehouse(cap, nz_cap)
        return a

    @classmethod
    def from_reference(cls, compressed: list, which: str, tokens: int, cap_tokens: Optional[int] = None,
                       headroom: float = DEFAULT_HEADROOM) -> "CompressedArena":
        """Re-house a reference-layout cache `[bitmaps, idxs, nzs, nz_offset]` holding `tokens` tokens per head."""
        bmp, idx, nzs, _ = compressed
        heads = len(nzs)
        t = tokens * cls.TILES_PER_TOKEN
        used = torch.tensor([n.numel() for n in nzs], dtype=torch.int64)
        cap = _round_up(cap_tokens if cap_tokens else tokens + DEFAULT_EXTRA_TOKENS, 256)
        per_token = float(used.max()) / max(tokens, 1)
        a = cls(heads, which, bmp.device, cap, _round_up(int(per_token * cap * headroom) + 1024, 8))
        a.bmp[:, :t] = bmp.view(heads, t)
        a.idx[:, :t + 1] = idx.view(heads, t + 1)
        for h in range(heads):
            a.nz[h, :nzs[h].numel()] = nzs[h]
        a.used, a.tokens = used, tokens
        return a

    def to_reference(self) -> list:
        """[bitmaps int64 [B', 2T], idxs int32 [B', 2T+1], list of B' fp16 streams, nz_offset] (contiguous copies)."""
        from .hook import FlatStreams, nz_offset_from_idxs
        t = self.tokens * self.TILES_PER_TOKEN
        bmp = self.bmp[:, :t].contiguous()
        idx = self.idx[:, :t + 1].contiguous()
        per_head: List[torch.Tensor] = [self.nz[h, :int(self.used[h])].clone() for h in range(self.heads)]
        return [bmp, idx, FlatStreams(per_head), nz_offset_from_idxs(idx, self.heads)]


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m
