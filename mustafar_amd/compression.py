"""Drop-in for the reference's `kernel/compression.py` host API (convert_key_batched / convert_value_batched)
plus the magnitude prune that precedes it in the model (dh_prune_key / dh_prune_value).

Same arguments, same return types as kernel/compression.py:249-339 and :341-432:
    (bitmaps int64 [B', t*D/64], accum_counts int32 [B', t*D/64 + 1], list of B' fp16 1-D tensors)
computed by the HIP kernels in csrc/compress.hip behind the C ABI (include/mustafar_hip.h).  One small
device->host read (B'+1 int64 offsets) remains because the packed sizes define the shapes of the returned
tensors; the reference needs 1 + 2B' `.item()` syncs for the same reason (:308, :333-334).  Two passes over the
rows with the read between them (default below 768 Ki rows); round 5 added the form that reads the rows ONCE (the one-pass compression
launch into worst-case regions, the read behind it, a copy launch that packs the streams into the exact-size buffer): the default from
768 Ki rows on (MUSTAFAR_CONVERT=onepass / twopass force one).  Round 6: the host does not copy the offsets back, it polls a
mirror of them in pinned memory (`_offsets_mirror`, `_await_offsets`).
The per-head tensors are views of one packed buffer (the reference clones each slice, :335) and remember it: they are
`StreamPiece`s, a tensor subclass whose only behaviour is that
    torch.cat(list of all the pieces of one buffer, in order)      (model :274, :314: once per layer and decode step)
returns that buffer itself instead of copying every head's stream again, and that the per-head concatenation of the trigger
    [torch.cat([old[b], new[b]], dim=0) for b in range(heads)]      (model :368, :390)
lands, head by head, in ONE new buffer, so that the next `torch.cat` of the resulting list is free again.  ALIASING: the result of
the first form IS the cache's buffer, not a copy (an in-place op on it would change the cache; the reference hook only reads it), and
a per-head concatenation asked for twice returns a fresh tensor the second time.  Everything else a
piece is asked to do, it does as the plain tensor it is.  The hook's source is untouched by this: it is the tensors it gets back
from `convert_*_batched` that know where they live.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import _lib


class _Backing:
    """One packed buffer and the pieces it is cut into: offs[b] .. offs[b + 1] (halfs) is head b's stream."""
    __slots__ = ("buf", "offs", "indices", "succ", "filled", "__weakref__")

    def __init__(self, buf: torch.Tensor, offs: List[int]):
        self.buf, self.offs = buf, offs
        self.indices = list(range(len(offs) - 1))
        self.succ = None     # (id of the other backing, backing that takes old + new per head): the trigger's concatenation target
        self.filled = set()  # heads of THIS buffer already written by a trigger's per-head concatenation (see _append_piece)


class StreamPiece(torch.Tensor):
    """A head's packed stream: a view into a `_Backing` buffer (`_bk`, `_ix`).  See the module docstring."""

    @staticmethod
    def wrap(view: torch.Tensor, bk: _Backing, ix: int) -> "StreamPiece":
        p = view.as_subclass(StreamPiece)
        p._bk, p._ix = bk, ix
        return p

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.cat and args and isinstance(args[0], (list, tuple)) and "out" not in kwargs:
            tensors = args[0]
            dim = kwargs.get("dim", args[1] if len(args) > 1 else 0)
            n = len(tensors)
            if dim == 0 and n and all(type(t) is StreamPiece and hasattr(t, "_bk") for t in tensors):   # (a piece made by anything but `pieces_of` / `wrap` knows no buffer: plain cat)
                bk = tensors[0]._bk
                if n == len(bk.indices) and [t._ix for t in tensors] == bk.indices and all(t._bk is bk for t in tensors):
                    return bk.buf                                       # every piece of one buffer, in order: the buffer
                if n == 2 and tensors[0]._ix == tensors[1]._ix and tensors[0]._bk is not tensors[1]._bk:
                    return _append_piece(tensors[0], tensors[1])        # the trigger's per-head concatenation
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def _append_piece(old: StreamPiece, new: StreamPiece) -> torch.Tensor:
    """torch.cat([old, new]) of head b's old stream and its newly compressed tokens, written into the buffer that will hold
    every head's concatenation (allocated when the first head asks); returns the view of it (a piece of the new buffer)."""
    ob, nb, b = old._bk, new._bk, old._ix
    if len(ob.offs) != len(nb.offs):
        with torch._C.DisableTorchFunctionSubclass():
            return torch.cat([old, new], dim=0)
    if ob.succ is None or ob.succ[0] is not nb:
        offs = [0]
        for i in range(len(ob.offs) - 1):
            offs.append(offs[-1] + (ob.offs[i + 1] - ob.offs[i]) + (nb.offs[i + 1] - nb.offs[i]))
        ob.succ = (nb, _Backing(torch.empty((offs[-1],), dtype=ob.buf.dtype, device=ob.buf.device), offs))
    tgt = ob.succ[1]
    if b in tgt.filled:
        # the same concatenation asked for a second time: a fresh tensor, as torch.cat promises -- the slot of the shared buffer
        # already belongs to the first result (which aliases the cache: INTEGRATION.md, "aliasing")
        with torch._C.DisableTorchFunctionSubclass():
            return torch.cat([old.as_subclass(torch.Tensor), new.as_subclass(torch.Tensor)], dim=0)
    tgt.filled.add(b)
    dst = tgt.buf[tgt.offs[b]:tgt.offs[b + 1]]
    with torch._C.DisableTorchFunctionSubclass():
        torch.cat([old.as_subclass(torch.Tensor), new.as_subclass(torch.Tensor)], dim=0, out=dst)
    return StreamPiece.wrap(dst, tgt, b)


def pieces_of(flat: torch.Tensor, offs: List[int]) -> List[torch.Tensor]:
    """Cut a packed buffer into per-head `StreamPiece`s (offs: B' + 1 boundaries in halfs).  At c3 (64 heads) this list was half of what a
    conversion call costs on the host (round 5: 2.3 -> 1.3 us per piece).  Round 6: ONE split call that does not record the views for autograd
    (`unsafe_split_with_sizes`: fp16 cache streams carry no gradient) and the fresh views RE-TYPED in place (`__class__` assignment: `StreamPiece`
    adds no storage to `torch.Tensor`, so CPython allows it) instead of a second tensor object per piece through `_make_subclass`: ~0.4 us per piece."""
    bk = _Backing(flat, list(offs))
    n = len(offs) - 1
    if n == 0:
        return []
    if offs[0] != 0 or offs[-1] != flat.numel():
        views = [flat[offs[b]:offs[b + 1]] for b in range(n)]
    else:
        views = list(torch.unsafe_split_with_sizes(flat, [offs[b + 1] - offs[b] for b in range(n)]))
    retype = _RETYPE_IN_PLACE
    for b, p in enumerate(views):
        if retype and type(p) is torch.Tensor:
            p.__class__ = StreamPiece
        else:                                   # `flat` itself a subclass (leave its type system alone), or a PyTorch whose tensor objects cannot be re-typed
            p = views[b] = torch.Tensor._make_subclass(StreamPiece, p)
        p._bk, p._ix = bk, b
    return views


def _can_retype_in_place() -> bool:
    """Whether this PyTorch lets a plain tensor OBJECT become a `StreamPiece` by `__class__` assignment (CPython allows it between heap types of one
    layout; true for every PyTorch 2.x tried).  Probed once at import, on a CPU scalar: where it is refused, `pieces_of` wraps every view instead."""
    try:
        t = torch.empty(0)
        t.__class__ = StreamPiece
        return type(t) is StreamPiece and (t + 1).numel() == 0
    except Exception:
        return False


_RETYPE_IN_PLACE = _can_retype_in_place()


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def kth_from_sparsity(target_sparsity: float, D: int) -> int:
    """`num_to_keep = max(1, int(target_sparsity * D))` -- models/llama_mustafar_kernel.py:97 (a k-th-smallest index)."""
    return max(1, int(target_sparsity * D))


def prune_magnitude(x: torch.Tensor, target_sparsity: float, out: torch.Tensor | None = None) -> torch.Tensor:
    """Per-token magnitude prune along the last dim (dh_prune_key / dh_prune_value, model :77-153)."""
    assert 0 <= target_sparsity < 1, "Target sparsity must be between 0 and 1"
    if not x.is_cuda or x.dtype != torch.float16:
        raise RuntimeError("prune_magnitude expects a float16 tensor on the GPU")
    D = x.shape[-1]
    if D != 128:
        raise RuntimeError("prune_magnitude: this build supports head_dim == 128")
    xc = x.contiguous()
    if out is None:
        out = torch.empty_like(xc)
    elif not (out.is_contiguous() and out.shape == xc.shape and out.dtype == xc.dtype and out.device == xc.device):
        raise RuntimeError("prune_magnitude: `out` must be a contiguous tensor like the input")
    L = _lib.load()
    with torch.cuda.device(x.device):
        err = L.mustafar_prune_magnitude(_stream_ptr(x.device), xc.data_ptr(), out.data_ptr(), xc.numel() // D, D,
                                         kth_from_sparsity(target_sparsity, D))
    _lib.check(err, "mustafar_prune_magnitude")
    return out.view(x.shape)


import threading as _threading

_mirrors = _threading.local()   # per host thread: {(device index, B'): (pinned int64 [B' + 1], its numpy view)}


def _offsets_mirror(dev: torch.device, B: int):
    """Pinned (device-visible) host memory the bitmap launch mirrors head_off into.  One per thread, device and head count, reused from call to
    call: only the launch of the CURRENT call writes it, and that call does not return before it has seen every entry."""
    pool = _mirrors.__dict__.setdefault("pool", {})
    key = (dev.index, B)
    m = pool.get(key)
    if m is None:
        pin = torch.empty((B + 1,), dtype=torch.int64, pin_memory=True)
        m = pool[key] = (pin, pin.numpy())
    return m


def _await_offsets(arr, stream: torch.cuda.Stream) -> List[int]:
    """Spin until no entry of the mirror is the negative sentinel.  Bounded by the stream: once it has drained every store has landed (a
    kernel's stores to host memory are complete when it is), so entries still negative then are an error, not a wait."""
    spins = 0
    while arr.min() < 0:
        spins += 1
        if spins % 256 == 0 and stream.query():
            if arr.min() < 0:
                raise RuntimeError("mustafar: the stream drained and the compression launch left no stream offsets behind")
            break
    return arr.tolist()


def _convert_twopass(x: torch.Tensor, which: str) -> Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor]]:
    """Round 1-4 form: bitmaps + offsets (first pass over the rows), host read of the sizes, packed streams (second pass).
    Round 6: the sizes reach the host through a mirror in pinned memory the host polls (`mustafar_compress_bitmap_mirrored`) instead of a
    device-to-host copy behind the stream: the second pass is queued ~15 us earlier.  MUSTAFAR_CONVERT_SYNC=copy keeps the copy."""
    B, M, N = x.shape
    dev = x.device
    tiles = M * N // 64
    bitmaps = torch.empty((B, tiles), dtype=torch.int64, device=dev)
    accum = torch.empty((B, tiles + 1), dtype=torch.int32, device=dev)
    head_off = torch.empty((B + 1,), dtype=torch.int64, device=dev)
    L = _lib.load()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        st = stream.cuda_stream
        if _CONVERT_SYNC_COPY:
            err = getattr(L, f"mustafar_compress_bitmap_{which}")(st, x.data_ptr(), B, M, N, bitmaps.data_ptr(),
                                                                  accum.data_ptr(), head_off.data_ptr())
            _lib.check(err, f"mustafar_compress_bitmap_{which}")
            offs = head_off.cpu().tolist()   # the one host sync: sizes of the returned tensors
        else:
            pin, arr = _offsets_mirror(dev, B)
            arr.fill(-1)
            err = L.mustafar_compress_bitmap_mirrored(st, x.data_ptr(), B, M, N, 1 if which == "key" else 0, bitmaps.data_ptr(),
                                                      accum.data_ptr(), head_off.data_ptr(), pin.data_ptr())
            _lib.check(err, "mustafar_compress_bitmap_mirrored")
            offs = _await_offsets(arr, stream)   # the one host wait: sizes of the returned tensors
        packed = torch.empty((offs[-1],), dtype=torch.float16, device=dev)
        err = getattr(L, f"mustafar_compress_pack_{which}")(st, x.data_ptr(), B, M, N, bitmaps.data_ptr(),
                                                            accum.data_ptr(), head_off.data_ptr(),
                                                            packed.data_ptr() if offs[-1] else None)
        _lib.check(err, f"mustafar_compress_pack_{which}")
    return bitmaps, accum, pieces_of(packed, offs)


import os as _os

# Which form a conversion call takes.  MEASURED (round 6, wall time of a call in us, two passes / one pass, key and value; tools/probes/convert_breakdown.py,
# profiles/r06_convert_breakdown.txt): c1 59 / 68 and 65 / 62, c2 56 / 76 and 54 / 67, c3 (64 heads x 7936 tokens) 99 / 114 and 101 / 106, c4 (32 x 32512)
# 168 / 153 and 238 / 139, c5 (128 x 16128) 330 / 284 and 472 / 279.  A call is device time + one host wait: the two-pass form has LESS device work in front
# of the wait (35 us of counting against 65 us of counting + packing at c3), the one-pass form reads the rows once and its copy launch beats the second pass
# from ~1 M rows on (the second pass of V runs at 2 TB/s there).  Default: two passes below 768 Ki rows, one pass from there on; MUSTAFAR_CONVERT=onepass /
# twopass selects one form for every size (equal results, tests/test_gpu_parity.py).  (Round 5, before the list of pieces was made cheap and the
# wait was a device-to-host copy: 188 / 215 at c3.)
_CONVERT_SYNC_COPY = _os.environ.get("MUSTAFAR_CONVERT_SYNC", "") == "copy"
_CONVERT_FORM = {"onepass": 1, "twopass": 2}.get(_os.environ.get("MUSTAFAR_CONVERT", ""), 0)
if _os.environ.get("MUSTAFAR_COMPRESS", "") == "twopass":
    _CONVERT_FORM = 2
_CONVERT_ONEPASS_ROWS = 768 * 1024
convert_fallbacks = 0   # conversions repeated through the two-pass form because a block of the one-pass launch gave up waiting (never seen)


def _convert(inputs: torch.Tensor, which: str, onepass: bool | None = None) -> Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor]]:
    """`onepass` (or MUSTAFAR_CONVERT=onepass; round 5): ONE read of the rows.  The one-pass compression launch writes bitmaps, offsets and every head's stream (into regions of
    worst-case size), the sizes are read on the host BEHIND that work (the reference's return type needs them: compression.py:308),
    and one copy launch packs the regions into the exact-size buffer.  Rounds 1-4 read the rows twice with the host read in between."""
    global convert_fallbacks
    B, M, N = inputs.shape
    assert inputs.is_cuda
    assert inputs.dim() == 3
    assert M % 64 == 0
    if inputs.dtype != torch.float16 or N != 128:
        raise RuntimeError("convert_*_batched expects float16 [B', t, 128]")
    x = inputs.contiguous()
    if onepass is None:
        onepass = _CONVERT_FORM == 1 or (_CONVERT_FORM == 0 and B * M >= _CONVERT_ONEPASS_ROWS)
    if not onepass or B * M * (N // 8) > 0xffffffff:
        return _convert_twopass(x, which)
    dev = x.device
    tiles = M * N // 64
    bitmaps = torch.empty((B, tiles), dtype=torch.int64, device=dev)
    accum = torch.empty((B, tiles + 1), dtype=torch.int32, device=dev)
    status = torch.zeros((B + 2,), dtype=torch.int64, device=dev)             # [head_off (B' + 1) | flag]
    L = _lib.load()
    st = _stream_ptr(dev)
    with torch.cuda.device(dev):
        regions = torch.empty((B * M * N,), dtype=torch.float16, device=dev)    # worst case: nothing pruned
        scratch = torch.empty((int(L.mustafar_convert_scratch_bytes(B, M)),), dtype=torch.uint8, device=dev)
        if _CONVERT_SYNC_COPY:
            err = L.mustafar_convert_onepass(st, x.data_ptr(), B, M, N, 1 if which == "key" else 0, bitmaps.data_ptr(), accum.data_ptr(),
                                             status.data_ptr(), regions.data_ptr(), status.data_ptr() + 8 * (B + 1), scratch.data_ptr())
            _lib.check(err, "mustafar_convert_onepass")
            host = status.cpu().tolist()     # the one host sync: sizes of the returned tensors (and the flag)
        else:
            pin, arr = _offsets_mirror(dev, B + 1)
            arr.fill(-1)
            err = L.mustafar_convert_onepass_mirrored(st, x.data_ptr(), B, M, N, 1 if which == "key" else 0, bitmaps.data_ptr(), accum.data_ptr(),
                                                      status.data_ptr(), regions.data_ptr(), status.data_ptr() + 8 * (B + 1), scratch.data_ptr(),
                                                      pin.data_ptr())
            _lib.check(err, "mustafar_convert_onepass_mirrored")
            host = _await_offsets(arr, torch.cuda.current_stream(dev))     # the one host wait: sizes of the returned tensors (and the flag)
        if host[B + 1] & 0xffffffff:
            convert_fallbacks += 1
            return _convert_twopass(x, which)
        offs = host[:B + 1]
        packed = torch.empty((offs[-1],), dtype=torch.float16, device=dev)
        err = L.mustafar_convert_pack(st, regions.data_ptr(), B, M, N, status.data_ptr(), packed.data_ptr() if offs[-1] else None)
        _lib.check(err, "mustafar_convert_pack")
    return bitmaps, accum, pieces_of(packed, offs)


def convert_key_batched(inputs: torch.Tensor):
    """kernel/compression.py:249-339.  inputs: pruned K [B', t, 128] fp16 (t % 64 == 0)."""
    return _convert(inputs, "key")


def convert_value_batched(inputs: torch.Tensor):
    """kernel/compression.py:341-432.  inputs: pruned V [B', t, 128] fp16 (t % 64 == 0)."""
    return _convert(inputs, "value")
