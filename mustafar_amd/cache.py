"""Device-resident compressed cache with in-place append.

The reference keeps the compressed cache as `[bitmaps, idxs, nzs(list per kv-head), nz_offset]` and, on every 256-token
trigger, rebuilds all of it: `torch.cat` of the bitmaps and offsets of every head, one `torch.cat` per head for the
streams, a Python list of device scalars for the offsets (models/llama_mustafar_kernel.py:339-390).  `CompressedArena`
holds the same four arrays with a little spare capacity -- a head's bitmap / offset rows are `cap_tokens` long, every head's
stream has its own region of `nz_cap` halfs -- so an append writes the NEW tokens only.

The format is unchanged (bit order, padding to 8, offsets in half2 units, `nz_offset` in uint4 units): the SpMV kernels
read an arena through `mustafar_cache_view` (head strides), and `to_reference()` returns the reference's contiguous list.

Sizing (round 3).  Whatever is allocated beyond the bytes in use counts against the metric's second half (peak KV bytes), so
an arena is housed at `(1 + slack)` x what it holds (`DEFAULT_SLACK` = 3 %: token rows and stream regions alike; at c3 that keeps dense / reserved bytes at 2.0) and nothing
is reserved for a worst-case append any more:
  * an append first makes room for what it EXPECTS to add (the head's measured halfs per token x 1.04) -- re-housing the arena
    at `(1 + slack)` x the new size when that does not fit (up to ~8 k tokens that is every trigger: one copy of the cache,
    what the reference does with a dozen `torch.cat`s per trigger anyway; at 16 k tokens every other trigger, at 32 k every fourth);
  * the launch reports every head's true new length and refuses to write past a region (device flag, bit 0); the host reads
    the flag and the lengths right behind the launch (one small device->host read per trigger, as `append()` always had) and, if
    a head did outgrow its region -- rows full of ties keep more than the expected count, model :107 -- re-houses at the
    measured size and repeats the launch (the raw rows are still in place: the window slides afterwards);
  * only when a region already has room for a worst-case append (t x 128 halfs: nothing can overflow) does the call stay
    asynchronous, as in round 2: lengths and flag then travel through pinned memory and are looked at on the next use.
Bit 1 of the flag (a block of the one-pass compression gave up waiting for the lengths in front of it: it relies on lower
workgroup ids being dispatched first) is a hard error with its own exception; it is never answered by a retry.

Growth by extents (round 3b).  Re-housing copies the whole cache, and at 8 k tokens a 3 % margin is smaller than one trigger: every
trigger paid one copy of the cache, a round of driver allocations and -- because every array moved -- a new capture of the decode
graph (c3: 16 + 11 ms per trigger against 1.4 ms per step).  `append_extent_pair` instead compresses the 256 tokens of a trigger into
a small arena of their own, an EXTENT, and lists its view in a device table next to the base arena (`ext_table`).  Nothing that is
already compressed is read, written or moved; the pair form of the one-pass decode launch takes the blocks behind the base tokens
from the table (mustafar_decode_attention_extents), so the base pointers, the table pointer and therefore a graph captured ahead
stay valid across the trigger.  `tokens` / the four arrays keep describing the BASE; `total_tokens` counts the extents too,
`to_reference()` concatenates, and after `MAX_EXTENTS` triggers (512: 128 k generated tokens) or when a launch form that cannot read extents is asked for
`consolidate()` re-houses everything into one base again.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import List, Optional

import torch

from . import _lib

DEFAULT_SLACK = 0.03


class ExtentPool:
    """ONE allocation holding the extents of one trigger of every layer (2 per layer: K, V -- all of one geometry: 256 tokens, the
    largest region any layer expects), their status words ([flag, K lengths, V lengths] per layer) and the compression scratch;
    initialised by four launches.  The extents are `CompressedArena`s carved from it (they keep the allocation alive)."""

    def __init__(self, n_layers: int, heads: int, device, nz_cap: int):
        self.n, self.heads, self.device, self.nz_cap, self.used = n_layers, heads, device, nz_cap, False
        offs, ext_bytes = CompressedArena._layout(heads, 256, nz_cap)
        self.ext_bytes, self._offs = ext_bytes, offs
        ne = 2 * n_layers
        self.status_words = 1 + 2 * heads                                  # int64 per layer: flag (low 4 bytes), K totals, V totals
        status_bytes = _round_up(n_layers * self.status_words * 8, 256)
        scratch_bytes = _round_up(n_layers * int(_lib.load().mustafar_compress_scratch_bytes(heads, 256)), 256)
        self.buf = torch.empty(ne * ext_bytes + status_bytes + scratch_bytes, dtype=torch.uint8, device=device)
        self.status = self.buf[ne * ext_bytes:ne * ext_bytes + n_layers * self.status_words * 8].view(torch.int64).view(n_layers, self.status_words)
        self.scratch = self.buf[ne * ext_bytes + status_bytes:]
        tiles = 256 * CompressedArena.TILES_PER_TOKEN
        w = self.buf[:ne * ext_bytes].view(torch.int32).view(ne, ext_bytes // 4)
        w[:, offs["nz_offset"] // 4:offs["nz_offset"] // 4 + heads] = _head_index(heads, device) * (nz_cap // 8)      # stream starts of the heads
        torch.as_strided(w, (ne, heads), (ext_bytes // 4, tiles + 1), offs["idx"] // 4).zero_()                         # offset 0 of every head
        w[:, offs["totals"] // 4:offs["flag"] // 4 + 1].zero_()                                                          # the extents' OWN lengths / flag words (an in-place append into one would read them)
        self.status.zero_()                                                                                              # flags (and lengths)
        self._host = torch.empty((n_layers, self.status_words), dtype=torch.int64).pin_memory()                         # landing area of read_status()

    def extent(self, j: int, which: str, slack: float) -> "CompressedArena":
        return CompressedArena(self.heads, which, self.device, 256, self.nz_cap, slack, storage=self.buf[j * self.ext_bytes:(j + 1) * self.ext_bytes])

    def totals_ptr(self, j: int) -> int:       # extent j = 2 * layer + side
        return self.status.data_ptr() + ((j // 2) * self.status_words + 1 + (j % 2) * self.heads) * 8

    def flag_ptr(self, layer: int) -> int:
        return self.status.data_ptr() + layer * self.status_words * 8

    def read_status(self) -> torch.Tensor:
        self._host.copy_(self.status, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self._host


compress_fallbacks = 0   # appends repeated in the two-pass form because a block of the one-pass launch gave up waiting (round 5; never seen outside tests)


class ArenaAppendTimeout(RuntimeError):
    """The one-pass compression launch gave up waiting for a predecessor block's length (device flag bit 1).  The appended
    tokens are incomplete; MUSTAFAR_COMPRESS=twopass selects the form without that dependence."""


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _cap_rows(tokens: int, slack: float) -> int:
    return _round_up(int(math.ceil(tokens * (1.0 + slack))), 64)


def _cap_nz(halfs: int, slack: float) -> int:
    return _round_up(int(halfs * (1.0 + slack)) + 256, 8)


_HEAD_INDEX = {}


def _head_index(heads: int, device) -> torch.Tensor:
    """arange(heads) as int32 on the device, kept (an arena per trigger and side is initialised from it)."""
    key = (heads, str(device))
    t = _HEAD_INDEX.get(key)
    if t is None:
        t = _HEAD_INDEX[key] = torch.arange(heads, dtype=torch.int32, device=device)
    return t


class CompressedArena:
    TILES_PER_TOKEN = 2   # head_dim 128 / 64
    MAX_EXTENTS = 512     # appended 256-token extents listed in the device table before a consolidation: 128 k tokens of generation, i.e. every
                          # length the one-pass launch serves (round 3: 64 -- a 170-ms copy of the cache every 16 k tokens at c3; the table is 28 KB)
    VIEW_BYTES = ctypes.sizeof(_lib.CacheView)

    def __init__(self, heads: int, which: str, device, cap_tokens: int, nz_cap: int, slack: float = DEFAULT_SLACK,
                 storage: Optional[torch.Tensor] = None):
        """storage: a uint8 tensor of `_layout(...)[1]` bytes to carve the arena from (an extent of a pooled trigger: the caller
        has initialised it -- nz_offset, idx[:, 0], flag -- for all its extents at once); None: an allocation of its own."""
        assert which in ("key", "value") and cap_tokens % 64 == 0 and nz_cap % 8 == 0
        self.heads, self.which, self.device, self.slack = heads, which, device, slack
        self.tokens = 0
        self.extents: List["CompressedArena"] = []   # appended 256-token extents, oldest first (module docstring)
        self._ext_table = None                        # device copy of their views (MAX_EXTENTS x mustafar_cache_view)
        self._alloc(cap_tokens, nz_cap, storage)
        if storage is None:
            self._init_empty()

    # ---- storage ---------------------------------------------------------------------------------------------
    @classmethod
    def _layout(cls, heads: int, cap_tokens: int, nz_cap: int):
        """Byte offsets of the arrays inside an arena's one allocation, and its size."""
        tiles = cap_tokens * cls.TILES_PER_TOKEN
        sizes = (("bmp", heads * tiles * 8), ("nz", heads * nz_cap * 2), ("idx", heads * (tiles + 1) * 4), ("nz_offset", heads * 4),
                 ("totals", heads * 8), ("flag", 4))
        offs, total = {}, 0
        for name, n in sizes:
            offs[name] = total
            total = _round_up(total + n, 256)
        return offs, total

    def _alloc(self, cap_tokens: int, nz_cap: int, storage: Optional[torch.Tensor] = None):
        """ONE device allocation per arena, carved into the four arrays of the format (+ the per-head lengths and the flag the
        append launch writes): a re-housing is one allocation and one copy launch (mustafar_cache_rehouse)."""
        tiles = cap_tokens * self.TILES_PER_TOKEN
        self.cap_tokens, self.nz_cap = cap_tokens, nz_cap
        H = self.heads
        offs, total = self._layout(H, cap_tokens, nz_cap)
        if storage is not None:
            assert storage.dtype == torch.uint8 and storage.numel() == total and storage.is_contiguous()
        self._buf = buf = storage if storage is not None else torch.empty(total, dtype=torch.uint8, device=self.device)
        cut = lambda name, n, dt: buf[offs[name]:offs[name] + n].view(dt)
        self.bmp = cut("bmp", H * tiles * 8, torch.int64).view(H, tiles)
        self.idx = cut("idx", H * (tiles + 1) * 4, torch.int32).view(H, tiles + 1)
        self.nz = cut("nz", H * nz_cap * 2, torch.float16).view(H, nz_cap)
        self.nz_offset = cut("nz_offset", H * 4, torch.int32)
        self._totals = cut("totals", H * 8, torch.int64)
        self._overflow = cut("flag", 4, torch.int32)
        self._used = torch.zeros(H, dtype=torch.int64)   # host copy of every head's stream length (halfs)
        self._host_totals = None                           # pinned landing area of an asynchronous append's lengths
        self._pending = None                               # (event, tokens appended) behind an asynchronous append
        self._host_flag = None
        self._view = _lib.CacheView(self.bmp.data_ptr(), self.nz.data_ptr(), self.idx.data_ptr(), self.nz_offset.data_ptr(),
                                    tiles, tiles + 1,
                                    nz_cap // 8 if (H * (nz_cap // 8) < 2 ** 32 and os.environ.get("MUSTAFAR_NZ_STRIDE", "1") != "0") else 0)

    def _init_empty(self):
        """An arena that starts empty: stream starts of the heads, offset 0 of every head, flag 0."""
        torch.mul(_head_index(self.heads, self.device), self.nz_cap // 8, out=self.nz_offset)
        self.idx[:, 0] = 0
        self._overflow.zero_()

    @property
    def view(self) -> "_lib.CacheView":
        return self._view

    def _settle(self, wait: bool = True) -> None:
        """Take in the lengths and the flag of an asynchronous append (see the module docstring).  wait=False: only if the copy
        has landed already (the decode path calls this before every read of the arena: no host stall, and a failed append
        is reported before the cache is used again rather than 256 steps later).  Unlike the synchronous appends this path has NO
        repeat in the two-pass form: by the time its flag is read the raw rows have slid out of the window, so a timeout (bit 1)
        rolls the token count back and raises ArenaAppendTimeout."""
        if self._pending is None:
            return
        ev, t = self._pending
        if not wait and not ev.query():
            return
        ev.synchronize()
        self._pending = None
        flag = int(self._host_flag[0])
        if flag:
            self.tokens -= t                     # the append did not complete: the cache is what it was before it
            if self.which == "key":              # (the pair's flag lives in the K arena)
                self._overflow.zero_()
            if flag & 2:
                raise ArenaAppendTimeout("CompressedArena: a block of the one-pass compression timed out waiting for the stream lengths in front "
                                         "of it; the 256 tokens of this trigger were not appended (MUSTAFAR_COMPRESS=twopass avoids the wait)")
            raise RuntimeError("CompressedArena: a head outgrew a stream region that had room for a worst-case append: this is a bug")
        self._used = self._host_totals.clone()

    @property
    def used(self) -> torch.Tensor:
        """Exact stream length of every head in halfs (host tensor); waits for an asynchronous append still in flight."""
        self._settle()
        return self._used

    @used.setter
    def used(self, value: torch.Tensor):
        self._pending = None
        self._used = value

    def poll(self) -> None:
        self._settle(wait=False)
        for e in self.extents:                   # (an extent that took the asynchronous path reports through the same mechanism)
            e._settle(wait=False)

    def view_ptr(self):
        return ctypes.byref(self._view)

    def bytes_in_use(self) -> int:
        t = self.tokens * self.TILES_PER_TOKEN
        return self.heads * (t * 8 + (t + 1) * 4 + 4) + int(self.used.sum()) * 2 + sum(e.bytes_in_use() for e in self.extents)

    def bytes_reserved(self) -> int:
        own = sum(x.numel() * x.element_size() for x in (self.bmp, self.idx, self.nz, self.nz_offset))
        return own + sum(e.bytes_reserved() for e in self.extents) + (self._ext_table.numel() if self._ext_table is not None else 0)

    # ---- growth by extents (module docstring) ------------------------------------------------------------------
    @property
    def total_tokens(self) -> int:
        return self.tokens + 256 * len(self.extents)

    @property
    def ext_table(self) -> torch.Tensor:
        """Device table of the extents' views; created on first use (a decode graph captured AHEAD of a trigger names it before
        the first extent exists)."""
        if self._ext_table is None:
            self._ext_table = torch.zeros(self.MAX_EXTENTS * self.VIEW_BYTES, dtype=torch.uint8, device=self.device)
        return self._ext_table

    def signature(self) -> tuple:
        """Addresses a captured decode graph holds for this cache (extents are found through the table at run time)."""
        return (self.bmp.data_ptr(), self.nz.data_ptr(), self.idx.data_ptr(), self.tokens, self.ext_table.data_ptr())

    def _list_extent(self, ext: "CompressedArena") -> None:
        i = len(self.extents)
        raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.byref(ext._view), self.VIEW_BYTES)), dtype=torch.uint8)
        self.ext_table[i * self.VIEW_BYTES:(i + 1) * self.VIEW_BYTES].copy_(raw)   # (stream-ordered: in front of every later launch)
        self.extents.append(ext)

    @staticmethod
    def append_extent_pair(k_arena: "CompressedArena", v_arena: "CompressedArena", k_rows: torch.Tensor, v_rows: torch.Tensor,
                           kth_k: int, kth_v: int) -> None:
        """The 256-token trigger without touching what is compressed already: rows [0, 256) of the window buffers become an extent
        of each side (prune + compress in one launch: from_raw_pair), listed in the device tables."""
        if len(k_arena.extents) >= k_arena.MAX_EXTENTS or len(k_arena.extents) != len(v_arena.extents) or k_arena.tokens % 256:
            raise RuntimeError("append_extent_pair: extent table full (consolidate() first) or K / V out of step")
        # regions sized from what the base measured per token (+ 4 %): right the first time, no estimate to shrink afterwards (a head
        # that needs more -- rows full of ties -- is answered by append_window_pair's repeat at the measured size)
        ek = CompressedArena(k_arena.heads, "key", k_arena.device, 256, _cap_nz(k_arena._expected_append(256, kth_k), 0.0), k_arena.slack)
        ev = CompressedArena(v_arena.heads, "value", v_arena.device, 256, _cap_nz(v_arena._expected_append(256, kth_v), 0.0), v_arena.slack)
        ek._landing, ek._blk_scratch = k_arena._host_landing(), k_arena._scratch_for(256)   # (the base's pinned landing area and scratch serve its extents)
        CompressedArena.append_window_pair(ek, ev, k_rows, v_rows, 256, kth_k, kth_v, expect=False)
        if ek._view.nz_head_stride == 0 or ev._view.nz_head_stride == 0:
            raise RuntimeError("append_extent_pair: extents need views with a stream stride")
        k_arena._list_extent(ek)
        v_arena._list_extent(ev)

    # ---- the trigger of ALL layers at once (round 4) ------------------------------------------------------------------------------
    @staticmethod
    def prepare_extents(pairs, kth_k: int, kth_v: int) -> "ExtentPool":
        """Storage for the NEXT trigger of every layer, allocated and initialised ahead of it (the host has nothing else to do between
        graph replays): ONE allocation for the 2 x len(pairs) extents, their status words and the compression scratch; three small
        launches set every extent's stream starts, first offsets and flags.  `pairs`: [(k_arena, v_arena), ...], one per layer."""
        k0 = pairs[0][0]
        heads, dev = k0.heads, k0.device
        need = 0
        for ka, va in pairs:
            if ka.heads != heads or va.heads != heads:
                raise RuntimeError("prepare_extents: every layer must have the same number of kv-heads")
            need = max(need, ka._expected_append(256, kth_k), va._expected_append(256, kth_v))
        return ExtentPool(len(pairs), heads, dev, _cap_nz(need, 0.0))

    @staticmethod
    def append_extent_pairs(pairs, rows, kth_k: int, kth_v: int, window_len: int, pool: Optional["ExtentPool"] = None) -> None:
        """The 256-token trigger of every layer (model :324-398 runs it layer by layer inside the attention forward): rows [0, 256) of
        each layer's window buffers become an extent of its K and of its V arena, and both windows slide by 256 rows.
            pairs : [(k_arena, v_arena), ...]      rows : [(k_buf, v_buf), ...] the window buffers [B, Hkv, cap, 128], `window_len` rows valid
        The layers' compression launches are issued back to back by ONE library call into a pooled allocation, ONE copy + wait brings
        every layer's flag and lengths to the host, one more call lists the extents in the device tables and slides the windows
        (a launch per layer).  A layer whose head outgrew its region (rows full of ties) is redone on its own at the measured size --
        its raw rows are still in place: nothing slides before every flag has been seen."""
        n = len(pairs)
        if n == 0:
            return
        k0 = pairs[0][0]
        for (ka, va), (kr, vr) in zip(pairs, rows):
            if len(ka.extents) >= ka.MAX_EXTENTS or len(ka.extents) != len(va.extents) or ka.tokens % 256 or ka.tokens != va.tokens:
                raise RuntimeError("append_extent_pairs: extent table full (consolidate() first) or K / V out of step")
            if kr.shape != vr.shape or kr.dtype != torch.float16 or kr.dim() != 4 or kr.shape[0] * kr.shape[1] != ka.heads or kr.shape[3] != 128 \
                    or kr.shape[2] < window_len or window_len < 256 or not kr.is_contiguous() or not vr.is_contiguous() or kr.shape != rows[0][0].shape:
                raise RuntimeError("append_extent_pairs expects contiguous fp16 [B, Hkv, rows >= window_len >= 256, 128] buffers of one shape")
        # a pool prepared some steps ahead is reused only if it still fits: same layers / heads / device, unused, and regions at least
        # what every layer expects to append NOW (a short pool would only be caught by the overflow flag and a slow per-layer redo)
        if pool is not None:
            need = max(max(ka._expected_append(256, kth_k), va._expected_append(256, kth_v)) for ka, va in pairs)
            if pool.n != n or pool.heads != k0.heads or pool.used or pool.device != k0.device or pool.nz_cap < _cap_nz(need, 0.0):
                pool = None
        if pool is None:
            pool = CompressedArena.prepare_extents(pairs, kth_k, kth_v)
        pool.used = True
        L = _lib.load()
        dev = k0.device
        items = (_lib.TriggerItem * n)()
        made = []
        for i, ((ka, va), (kr, vr)) in enumerate(zip(pairs, rows)):
            ek, ev = pool.extent(2 * i, "key", ka.slack), pool.extent(2 * i + 1, "value", va.slack)
            ka.ext_table, va.ext_table
            it = items[i]
            it.k_window, it.v_window = kr.data_ptr(), vr.data_ptr()
            it.k_dst, it.v_dst = ek._view, ev._view
            it.k_table_slot = ka._ext_table.data_ptr() + len(ka.extents) * CompressedArena.VIEW_BYTES
            it.v_table_slot = va._ext_table.data_ptr() + len(va.extents) * CompressedArena.VIEW_BYTES
            it.k_head_total, it.v_head_total, it.overflow_flag = pool.totals_ptr(2 * i), pool.totals_ptr(2 * i + 1), pool.flag_ptr(i)
            made.append((ek, ev))
        head_stride = rows[0][0].shape[2] * 128
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(L.mustafar_trigger_compress_batch(st, n, items, head_stride, k0.heads, 256, 128, kth_k, kth_v, pool.nz_cap, pool.nz_cap,
                                                         pool.scratch.data_ptr()), "mustafar_trigger_compress_batch")
            host = pool.read_status()                 # ONE copy into pinned memory + ONE wait: [n pairs][flag, K totals, V totals]
            redo, timed_out = [], set()
            for i in range(n):
                flag = int(host[i, 0]) & 0xffffffff
                if flag:   # bit 0 (a head outgrew its region) or bit 1 (a block gave up waiting: round 5 repeats instead of raising): this layer
                    redo.append(i)   # is redone on its own below -- its raw rows are still in place, nothing has slid
                    if flag & 2:
                        timed_out.add(i)
            keep = [i for i in range(n) if i not in redo]
            if keep:
                sub = items if not redo else (_lib.TriggerItem * len(keep))(*[items[i] for i in keep])
                _lib.check(L.mustafar_trigger_finish_batch(st, len(keep), sub, head_stride, k0.heads, window_len, 256), "mustafar_trigger_finish_batch")
        H = k0.heads
        for i in keep:
            (ka, va), (ek, ev) = pairs[i], made[i]
            ek.used, ek.tokens = host[i, 1:1 + H].clone(), 256
            ev.used, ev.tokens = host[i, 1 + H:1 + 2 * H].clone(), 256
            ka.extents.append(ek)
            va.extents.append(ev)
        global compress_fallbacks
        for i in redo:                                # bit 0: on its own, at the size the launch reported (append_window_pair's repeat);
            (ka, va), (kr, vr) = pairs[i], rows[i]    # bit 1: on its own in the two-pass form, which has no wait between workgroups
            prev_form = L.mustafar_compress_get_form()   # (this thread's; restored, not reset: a caller may have chosen a form itself)
            if i in timed_out:
                compress_fallbacks += 1
                _lib.check(L.mustafar_compress_set_form(2), "mustafar_compress_set_form")
            try:
                CompressedArena.append_extent_pair(ka, va, kr, vr, kth_k, kth_v)
            finally:
                if i in timed_out:
                    L.mustafar_compress_set_form(prev_form)
            with torch.cuda.device(dev):
                _lib.check(L.mustafar_window_drop_front(torch.cuda.current_stream(dev).cuda_stream, kr.data_ptr(), vr.data_ptr(), head_stride,
                                                        H, window_len, 256), "mustafar_window_drop_front")

    def drop_extents(self) -> None:
        """Back to the base tokens (the extents never touched them); their table entries are overwritten by the next ones."""
        self.extents = []

    def consolidate(self) -> "CompressedArena":
        """One base arena holding everything (a copy of the cache; the addresses of the result are new).  Round 5: on the device -- two
        launches, no host read (mustafar_cache_rehouse for the base, mustafar_cache_consolidate_extents for the extents, found through
        the device table the decode launch reads): c3, 32 layers, K and V: 162 ms through the reference layout on the host -> a few ms."""
        if not self.extents:
            return self
        n = len(self.extents)
        if self._view.nz_head_stride == 0 or any(e._view.nz_head_stride == 0 for e in self.extents) or 3 * n > 65535 or \
                os.environ.get("MUSTAFAR_CONSOLIDATE", "") == "host":
            return CompressedArena.from_reference(self.to_reference(), self.which, self.total_tokens, None, self.slack)
        base_used = self.used.clone()                                  # (host tensors; resolves asynchronous appends still in flight)
        ext_used = [e.used for e in self.extents]
        used = base_used.clone()
        for u in ext_used:
            used += u
        total = self.total_tokens
        new = CompressedArena(self.heads, self.which, self.device, _cap_rows(total, self.slack), _cap_nz(int(used.max()), self.slack), self.slack)
        if int(used.max()) > new.nz_cap:   # (the device copy trusts these host-side lengths for the size of the regions)
            raise RuntimeError("CompressedArena.consolidate: a head's summed stream length exceeds the new region: this is a bug")
        L = _lib.load()
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(L.mustafar_cache_rehouse(st, ctypes.byref(self._view), ctypes.byref(new._view), self.heads, self.tokens,
                                                _round_up(int(base_used.max()) if self.tokens else 0, 8)), "mustafar_cache_rehouse")
            _lib.check(L.mustafar_cache_consolidate_extents(st, ctypes.byref(new._view), self.ext_table.data_ptr(), n, self.heads, self.tokens,
                                                            _round_up(max(int(u.max()) for u in ext_used), 8)), "mustafar_cache_consolidate_extents")
        new.used, new.tokens = used, total
        return new

    def _rehouse(self, cap_tokens: int, nz_cap: int):
        """Move the cache into rows of `cap_tokens` tokens and stream regions of `nz_cap` halfs (never below what it holds):
        one allocation, one launch (mustafar_cache_rehouse); the old storage is released when the copy has been enqueued."""
        used, tokens = self.used.clone(), self.tokens              # (resolves a pending asynchronous append)
        m = int(used.max()) if tokens else 0
        assert cap_tokens >= tokens and nz_cap >= m
        old_buf, old_view, old = self._buf, self._view, (self.bmp, self.idx, self.nz)   # (old_buf keeps the source alive across the launch)
        self._alloc(cap_tokens, nz_cap)
        if self._view.nz_head_stride != 0:
            with torch.cuda.device(self.device):
                err = _lib.load().mustafar_cache_rehouse(torch.cuda.current_stream(self.device).cuda_stream, ctypes.byref(old_view),
                                                         ctypes.byref(self._view), self.heads, tokens, _round_up(m, 8))
            _lib.check(err, "mustafar_cache_rehouse")
            self._overflow.zero_()
        else:   # (MUSTAFAR_NZ_STRIDE=0, an experiment switch: the views carry no stream stride; tensor copies instead)
            self._init_empty()
            t2 = tokens * self.TILES_PER_TOKEN
            self.bmp[:, :t2] = old[0][:, :t2]
            self.idx[:, :t2 + 1] = old[1][:, :t2 + 1]
            self.nz[:, :m] = old[2][:, :m]
        del old_buf, old
        self.used, self.tokens = used, tokens

    def _make_room(self, t: int, need_halfs: int) -> None:
        """Rows for t more tokens and stream regions of at least need_halfs, re-housed at (1 + slack) x the new size if either is short."""
        rows, nz = self.cap_tokens, self.nz_cap
        if self.tokens + t > rows:
            rows = _cap_rows(self.tokens + t, self.slack)
        if need_halfs > nz:
            nz = _cap_nz(need_halfs, self.slack)
        if rows != self.cap_tokens or nz != self.nz_cap:
            self._rehouse(rows, nz)

    def _expected_append(self, t: int, kth: int) -> int:
        """Halfs a t-token append is expected to add to the fullest head: its measured halfs per token (the kept count + padding
        to 8 per tile; + 4 %), or the count the prune rule keeps without ties when nothing has been measured yet."""
        used = int(self.used.max()) if self.tokens else 0
        if self.tokens >= 64 and used > 0:
            per_token = used / float(self.tokens)
        else:
            per_token = float(136 if kth <= 0 else 128 - kth + 1 + 8)
        return int(t * per_token * 1.04) + 256

    # ---- append of PRUNED tokens, two passes (model :339-390) --------------------------------------------------
    def append(self, x: torch.Tensor) -> None:
        """x: pruned fp16 [B', t, 128], t % 64 == 0 -> appended behind the tokens in use."""
        if x.dim() != 3 or x.shape[0] != self.heads or x.shape[2] != 128 or x.shape[1] % 64 or x.dtype != torch.float16:
            raise RuntimeError("CompressedArena.append expects a pruned fp16 [B', t, 128] tensor with t % 64 == 0")
        if not x.is_contiguous():
            x = x.contiguous()
        t = x.shape[1]
        L = _lib.load()
        self._make_room(t, 0)
        st = torch.cuda.current_stream(self.device).cuda_stream
        key = self.which == "key"
        with torch.cuda.device(self.device):
            f = L.mustafar_cache_append_bitmap_key if key else L.mustafar_cache_append_bitmap_value
            _lib.check(f(st, x.data_ptr(), self.heads, t, 128, self.view_ptr(), self.tokens, self._totals.data_ptr()),
                       "mustafar_cache_append_bitmap")
            totals = self._totals.cpu()                  # the one host read of an append (B' values)
            need = int(totals.max())
            if need > self.nz_cap:                       # the new tiles do not fit behind some head's stream: re-house, redo pass 1
                self._make_room(t, need)
                _lib.check(f(st, x.data_ptr(), self.heads, t, 128, self.view_ptr(), self.tokens, self._totals.data_ptr()),
                           "mustafar_cache_append_bitmap")
            g = L.mustafar_cache_append_pack_key if key else L.mustafar_cache_append_pack_value
            _lib.check(g(st, x.data_ptr(), self.heads, t, 128, self.view_ptr(), self.tokens), "mustafar_cache_append_pack")
        self.used, self.tokens = totals, self.tokens + t

    # ---- fused trigger: prune + compress + append of the RAW window rows of K and V (model :324-398) -----------------------
    @staticmethod
    def _launch_pair(k_arena, v_arena, k_rows, v_rows, t, kth_k, kth_v):
        L = _lib.load()
        dev = k_arena.device
        scratch = k_arena._scratch_for(t)
        flag = k_arena._overflow
        st = torch.cuda.current_stream(dev).cuda_stream
        err = L.mustafar_cache_append_kv(st, k_rows.data_ptr(), v_rows.data_ptr(), k_rows.shape[2] * 128, k_arena.heads, t, 128, kth_k, kth_v,
                                         k_arena.view_ptr(), v_arena.view_ptr(), k_arena.tokens, k_arena._totals.data_ptr(),
                                         v_arena._totals.data_ptr(), k_arena.nz_cap, v_arena.nz_cap, flag.data_ptr(), scratch.data_ptr())
        _lib.check(err, "mustafar_cache_append_kv")
        return flag

    @staticmethod
    def append_window_pair(k_arena: "CompressedArena", v_arena: "CompressedArena", k_rows: torch.Tensor, v_rows: torch.Tensor,
                           t: int, kth_k: int, kth_v: int, expect: bool = True) -> None:
        """k_rows / v_rows: fp16 [B, Hkv, >= t, 128] buffers (a window: rows [0, t) of every head are compressed; the row
        stride between heads is the buffer's) holding RAW (unpruned) tokens; kth = max(1, int(sparsity * 128)) (model :97),
        0 for rows that are already pruned.  One launch for both sides (mustafar_cache_append_kv), then -- unless both arenas had
        room for a worst-case append -- one read of the flag and the lengths, and a repeat at the measured size if a head
        outgrew its region (module docstring).  The rows must stay in place until this returns.  expect=False: the caller has
        just sized the regions itself (from_raw_pair): no room is made beforehand, an overflow is answered by the measured size."""
        global compress_fallbacks
        heads = k_arena.heads
        if v_arena.heads != heads or k_arena.tokens != v_arena.tokens or t % 64 or t <= 0:
            raise RuntimeError("append_window_pair: K and V arenas must describe the same heads and tokens; t % 64 == 0")
        if k_arena.extents or v_arena.extents:
            raise RuntimeError("append_window_pair: the cache has grown by extents; append_extent_pair() or consolidate()")
        for x in (k_rows, v_rows):
            if x.dtype != torch.float16 or x.dim() != 4 or x.shape[0] * x.shape[1] != heads or x.shape[3] != 128 or x.shape[2] < t \
                    or not x.is_contiguous():
                raise RuntimeError("append_window_pair expects contiguous fp16 [B, Hkv, rows >= t, 128] buffers")
        if k_rows.shape[2] != v_rows.shape[2]:
            raise RuntimeError("append_window_pair: K and V buffers must have the same number of rows per head")
        dev = k_arena.device
        for a, kth in ((k_arena, kth_k), (v_arena, kth_v)):
            used = int(a.used.max()) if a.tokens else 0            # (a.used waits for an asynchronous append still in flight)
            a._make_room(t, used + (a._expected_append(t, kth) if expect else 0))
        worst = t * 128
        safe = all(a.nz_cap - (int(a.used.max()) if a.tokens else 0) >= worst for a in (k_arena, v_arena))
        # (one flag per pair and call: K's.  It is NOT stored in the V arena: a view of K's allocation there kept K's old buffer alive
        # -- a whole extra K arena of memory -- from a re-housing of K until the next trigger)
        with torch.cuda.device(dev):
            if safe:   # nothing can overflow: stay asynchronous, lengths and flag through pinned memory (looked at on the next use)
                CompressedArena._launch_pair(k_arena, v_arena, k_rows, v_rows, t, kth_k, kth_v)
                ev = torch.cuda.Event()
                for a in (k_arena, v_arena):
                    if a._host_totals is None:
                        a._host_totals = torch.zeros(a.heads, dtype=torch.int64).pin_memory()
                        a._host_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
                    a._host_totals.copy_(a._totals, non_blocking=True)
                    a._host_flag.copy_(k_arena._overflow, non_blocking=True)
                ev.record(torch.cuda.current_stream(dev))
                for a in (k_arena, v_arena):
                    a._pending = (ev, t)
                    a.tokens += t
                return
            twopass = False
            attempt = 0
            while True:
                prev_form = _lib.load().mustafar_compress_get_form()   # (this thread's; restored, not reset)
                if twopass:
                    _lib.check(_lib.load().mustafar_compress_set_form(2), "mustafar_compress_set_form")
                try:
                    CompressedArena._launch_pair(k_arena, v_arena, k_rows, v_rows, t, kth_k, kth_v)
                finally:
                    if twopass:
                        _lib.load().mustafar_compress_set_form(prev_form)
                # flag and lengths: three small copies into pinned memory, ONE wait (for the launch and the copies)
                host = k_arena._host_landing()
                host[0][:1].copy_(k_arena._overflow, non_blocking=True)
                host[1].copy_(k_arena._totals, non_blocking=True)
                host[2].copy_(v_arena._totals, non_blocking=True)
                torch.cuda.current_stream(dev).synchronize()
                flag = int(host[0][0])
                totals = (host[1].clone(), host[2].clone())
                if flag:
                    k_arena._overflow.zero_()
                if flag & 2:
                    # a block of the one-pass launch gave up waiting for the lengths in front of it (never seen; the wait relies on in-order
                    # dispatch): the raw rows are still in place and the call is idempotent -- repeat it ONCE in the two-pass form, which has
                    # no such wait (round 5; round 4 raised here).  A second failure is an error.
                    if twopass:
                        raise ArenaAppendTimeout("CompressedArena: the append failed in the two-pass form as well (flag %d); nothing was appended" % flag)
                    compress_fallbacks += 1
                    twopass = True
                    continue
                if not flag:
                    for a, tot in zip((k_arena, v_arena), totals):
                        a.used, a.tokens = tot, a.tokens + t
                    return
                if attempt:
                    raise RuntimeError("CompressedArena: a head outgrew a stream region sized from the launch's own report: this is a bug")
                attempt += 1
                for a, tot in zip((k_arena, v_arena), totals):     # bit 0: the lengths the launch reported are exact: house them and repeat
                    a._make_room(t, int(tot.max()))

    def _host_landing(self):
        h = getattr(self, "_landing", None)
        if h is None:
            h = self._landing = (torch.zeros(2, dtype=torch.int32).pin_memory(), torch.zeros(self.heads, dtype=torch.int64).pin_memory(),
                                 torch.zeros(self.heads, dtype=torch.int64).pin_memory())
        return h

    def _scratch_for(self, t: int) -> torch.Tensor:
        n = int(_lib.load().mustafar_compress_scratch_bytes(self.heads, t))
        sc = getattr(self, "_blk_scratch", None)
        if sc is None or sc.numel() < n:
            self._blk_scratch = sc = torch.empty(n, dtype=torch.uint8, device=self.device)
        return sc

    @classmethod
    def from_raw_pair(cls, k_rows: torch.Tensor, v_rows: torch.Tensor, t: int, kth_k: int, kth_v: int, cap_tokens: Optional[int] = None,
                      slack: float = DEFAULT_SLACK):
        """Prefill (model :416-437): prune + compress the first t tokens of raw K / V [B, Hkv, L, 128] into two new arenas,
        one read of the dense block per side and no pruned copy.  Rows: (1 + slack) x t (at least `cap_tokens`); stream regions:
        from the count the prune rule keeps without ties (+ 8 halfs of padding per token, the average of two tiles rounded up to
        eight), grown to the measured size and the launch repeated if a head needs more, and shrunk if far too large."""
        heads = k_rows.shape[0] * k_rows.shape[1]
        rows = max(_cap_rows(t, slack), _round_up(cap_tokens or 0, 64))

        def region(kth: int) -> int:   # halfs per head
            kept = 128 if kth <= 0 else 128 - kth + 1
            return _cap_nz(int(t * min(kept + 8, 136) * 1.01) + 512, slack)

        k = cls(heads, "key", k_rows.device, rows, region(kth_k), slack)
        v = cls(heads, "value", v_rows.device, rows, region(kth_v), slack)
        cls.append_window_pair(k, v, k_rows, v_rows, t, kth_k, kth_v, expect=False)
        for a in (k, v):
            tight = _cap_nz(int(a.used.max()), slack)
            if a.nz_cap > tight + tight // 32:          # an estimate above the data (rows with many zeros): give the room back
                a._rehouse(a.cap_tokens, tight)
        return k, v

    # ---- conversion ------------------------------------------------------------------------------------------
    @classmethod
    def from_pruned(cls, x: torch.Tensor, which: str, cap_tokens: Optional[int] = None, slack: float = DEFAULT_SLACK) -> "CompressedArena":
        """Compress x [B', t, 128] (already pruned) into a new arena: rows (1 + slack) x t (at least `cap_tokens`), stream regions
        (1 + slack) x the measured halfs."""
        heads, t, _ = x.shape
        rows = max(_cap_rows(t, slack), _round_up(cap_tokens or 0, 64))
        a = cls(heads, which, x.device, rows, _round_up(t * 56 + 1024, 8), slack)   # (append() re-houses at the measured size if short)
        a.append(x)
        tight = _cap_nz(int(a.used.max()), slack)
        if a.nz_cap > tight + tight // 32:
            a._rehouse(a.cap_tokens, tight)
        return a

    @classmethod
    def from_reference(cls, compressed: list, which: str, tokens: int, cap_tokens: Optional[int] = None,
                       slack: float = DEFAULT_SLACK) -> "CompressedArena":
        """Re-house a reference-layout cache `[bitmaps, idxs, nzs, nz_offset]` holding `tokens` tokens per head."""
        bmp, idx, nzs, _ = compressed
        heads = len(nzs)
        t = tokens * cls.TILES_PER_TOKEN
        used = torch.tensor([n.numel() for n in nzs], dtype=torch.int64)
        rows = max(_cap_rows(tokens, slack), _round_up(cap_tokens or 0, 64))
        a = cls(heads, which, bmp.device, rows, _cap_nz(int(used.max()), slack), slack)
        a.bmp[:, :t] = bmp.view(heads, t)
        a.idx[:, :t + 1] = idx.view(heads, t + 1)
        for h in range(heads):
            a.nz[h, :nzs[h].numel()] = nzs[h]
        a.used, a.tokens = used, tokens
        return a

    def to_reference(self) -> list:
        """[bitmaps int64 [B', 2T], idxs int32 [B', 2T+1], list of B' fp16 streams, nz_offset] (contiguous copies)."""
        from .hook import append_compressed
        base = self._base_reference()
        if not self.extents:
            return base
        tokens = self.tokens   # base + extents, concatenated the way the model appends (llama_mustafar_kernel.py:339-390)
        for e in self.extents:
            base = append_compressed(base, e._base_reference(), self.heads, tokens, 256, 128)
            tokens += 256
        return [base[0].view(self.heads, -1), base[1].view(self.heads, -1), base[2], base[3]]

    def _base_reference(self) -> list:
        from .compression import pieces_of
        from .hook import FlatStreams, nz_offset_from_idxs
        t = self.tokens * self.TILES_PER_TOKEN
        bmp = self.bmp[:, :t].contiguous()
        idx = self.idx[:, :t + 1].contiguous()
        used = [int(u) for u in self.used]
        offs = [0]
        for u in used:
            offs.append(offs[-1] + u)
        flat = torch.empty((offs[-1],), dtype=torch.float16, device=self.device)
        for h, u in enumerate(used):
            flat[offs[h]:offs[h + 1]] = self.nz[h, :u]
        return [bmp, idx, FlatStreams(pieces_of(flat, offs)), nz_offset_from_idxs(idx, self.heads)]

