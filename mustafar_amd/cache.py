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
        self.used = torch.zeros(self.heads, dtype=torch.int64)   # host copy of every head's stream length (halfs)
        self._view = _lib.CacheView(self.bmp.data_ptr(), self.nz.data_ptr(), self.idx.data_ptr(), self.nz_offset.data_ptr(),
                                    tiles, tiles + 1)

    @property
    def view(self) -> "_lib.CacheView":
        return self._view

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
        old = (self.bmp, self.idx, self.nz, self.used.clone(), self.tokens)
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
