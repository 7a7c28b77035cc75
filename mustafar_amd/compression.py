"""Drop-in for the reference's `kernel/compression.py` host API (convert_key_batched / convert_value_batched)
plus the magnitude prune that precedes it in the model (dh_prune_key / dh_prune_value).

Same arguments, same return types as kernel/compression.py:249-339 and :341-432:
    (bitmaps int64 [B', t*D/64], accum_counts int32 [B', t*D/64 + 1], list of B' fp16 1-D tensors)
computed by the HIP kernels in csrc/compress.hip behind the C ABI (include/mustafar_hip.h).  One small
device->host read (B'+1 int64 offsets) sits between the two passes because the packed sizes define the
shapes of the returned tensors; the reference needs 1 + 2B' `.item()` syncs for the same reason (:308, :333-334).
The per-head tensors are views of one packed buffer (the reference clones each slice, :335).
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import _lib


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


def _convert(inputs: torch.Tensor, which: str) -> Tuple[torch.Tensor, torch.Tensor, List[torch.Tensor]]:
    B, M, N = inputs.shape
    assert inputs.is_cuda
    assert inputs.dim() == 3
    assert M % 64 == 0
    if inputs.dtype != torch.float16 or N != 128:
        raise RuntimeError("convert_*_batched expects float16 [B', t, 128]")
    x = inputs.contiguous()
    dev = x.device
    tiles = M * N // 64
    bitmaps = torch.empty((B, tiles), dtype=torch.int64, device=dev)
    accum = torch.empty((B, tiles + 1), dtype=torch.int32, device=dev)
    head_off = torch.empty((B + 1,), dtype=torch.int64, device=dev)
    L = _lib.load()
    st = _stream_ptr(dev)
    with torch.cuda.device(dev):
        err = getattr(L, f"mustafar_compress_bitmap_{which}")(st, x.data_ptr(), B, M, N, bitmaps.data_ptr(),
                                                              accum.data_ptr(), head_off.data_ptr())
        _lib.check(err, f"mustafar_compress_bitmap_{which}")
        offs = head_off.cpu().tolist()   # the one host sync: sizes of the returned tensors
        packed = torch.empty((offs[-1],), dtype=torch.float16, device=dev)
        err = getattr(L, f"mustafar_compress_pack_{which}")(st, x.data_ptr(), B, M, N, bitmaps.data_ptr(),
                                                            accum.data_ptr(), head_off.data_ptr(),
                                                            packed.data_ptr() if offs[-1] else None)
        _lib.check(err, f"mustafar_compress_pack_{which}")
    return bitmaps, accum, [packed[offs[b]:offs[b + 1]] for b in range(B)]


def convert_key_batched(inputs: torch.Tensor):
    """kernel/compression.py:249-339.  inputs: pruned K [B', t, 128] fp16 (t % 64 == 0)."""
    return _convert(inputs, "key")


def convert_value_batched(inputs: torch.Tensor):
    """kernel/compression.py:341-432.  inputs: pruned V [B', t, 128] fp16 (t % 64 == 0)."""
    return _convert(inputs, "value")
