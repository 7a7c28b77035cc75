"""Drop-in for the reference's `mustafar_package` PyTorch extension (kernel/kernel_wrapper/*).

Same two functions, same positional arguments, same dtype/device/contiguity checks and error types as
`mustafar_wrapper.cu:19-133` (key) and `:139-263` (value); the work is done by the HIP kernels behind the
C ABI in include/mustafar_hip.h (libmustafar_hip.so), launched on torch's current stream.  PyTorch is used
for device memory and the stream handle only.

Differences from the reference, all supersets:
  * `bmp` is consumed as int64 in place (the reference copies it to uint64 on every call, :90/:211);
  * the output is fully written by the kernel (the reference allocates zeros and writes on top);
  * `B` may carry 1 row instead of the 8 zero-padded rows the CUDA MMA tile needs -- the result then has 1 row
    ([Batch, 1, M]); with 8 rows the result is [Batch, 8, M] exactly as in the reference, every row computed;
  * the value path splits the token axis over workgroups (the reference's dormant Split_K) using an internal
    fp32 workspace; the caller's `Reduction_Workspace` (a 1-element tensor in the model) is accepted and unused;
  * shape arguments are validated (the reference validates nothing and silently launches nothing for N != 8).
"""
from __future__ import annotations

import torch

from . import _lib


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _common_checks(bmp, NZ, idx, NZ_Offset, B, require_B_contiguous: bool):
    # mustafar_wrapper.cu:36-38 / :156-158
    if B.device != bmp.device or B.device != NZ.device or B.device != idx.device or B.device != NZ_Offset.device:
        raise RuntimeError("All input tensors must be on the same device.")
    # :43-63 / :163-183
    if B.dtype != torch.float16:
        raise RuntimeError("Tensor B must be of type float16.")
    if NZ.dtype != torch.float16:
        raise RuntimeError("Tensor NZ must be of type float16.")
    if bmp.dtype != torch.int64:
        raise RuntimeError("Tensor bmp must be of type int64.")
    if idx.dtype != torch.int32:
        raise RuntimeError("Tensor idx must be of type int.")
    if NZ_Offset.dtype != torch.int32:
        raise RuntimeError("Tensor NZ_Offset must be of type int.")
    # :65-73 / :185-194 (TORCH_CHECK -> RuntimeError); the value path does not require B contiguous (:187)
    ok = bmp.is_contiguous() and NZ.is_contiguous() and idx.is_contiguous() and NZ_Offset.is_contiguous()
    if require_B_contiguous:
        ok = ok and B.is_contiguous()
    if not ok:
        raise RuntimeError("bmp, NZ, idx, B, C, and Reduction_Workspace tensors must be contiguous.")
    if not (bmp.is_cuda and NZ.is_cuda and idx.is_cuda and B.is_cuda and NZ_Offset.is_cuda):
        raise RuntimeError("bmp, NZ, idx, B, C, and (not)Reduction_Workspace tensors must be on CUDA device.")


def _rows(B: torch.Tensor, Batch_Size: int, inner: int) -> int:
    if Batch_Size <= 0 or inner <= 0 or B.numel() % (Batch_Size * inner):
        raise RuntimeError(f"Tensor B has {B.numel()} elements, not Batch_Size*N*{inner}")
    N = B.numel() // (Batch_Size * inner)
    if N not in (1, 8):
        raise RuntimeError(f"Tensor B must hold 1 or 8 rows per batch entry (got {N})")
    return N


def _check_cache_shapes(bmp, idx, NZ_Offset, tiles_per_head: int, Batch_Size: int, groups: int):
    if groups < 1 or Batch_Size % groups:
        raise RuntimeError("Batch_Size must be a multiple of num_key_value_groups")
    heads = Batch_Size // groups
    if bmp.numel() != heads * tiles_per_head or idx.numel() != heads * (tiles_per_head + 1) or NZ_Offset.numel() != heads:
        raise RuntimeError(
            f"compressed cache does not match the arguments: expected {heads} heads x {tiles_per_head} tiles "
            f"(bmp {bmp.numel()}, idx {idx.numel()}, NZ_Offset {NZ_Offset.numel()})")


def mustafar_key_formulation(bmp, NZ, idx, NZ_Offset, B, M_Global: int, K_Global: int, Batch_Size: int,
                             num_key_value_groups: int) -> torch.Tensor:
    """scores[b, n, m] = fp16(sum_k Khat[b // groups][m, k] * B[b, n, k])   (mustafar_wrapper.cu:19-133)."""
    _common_checks(bmp, NZ, idx, NZ_Offset, B, require_B_contiguous=True)
    if K_Global != 128 or M_Global <= 0 or M_Global % 64:
        raise RuntimeError("mustafar_key_formulation: need K_Global == 128 and M_Global a positive multiple of 64")
    N = _rows(B, Batch_Size, K_Global)
    _check_cache_shapes(bmp, idx, NZ_Offset, M_Global * K_Global // 64, Batch_Size, num_key_value_groups)
    L = _lib.load()
    C = torch.empty((Batch_Size, N, M_Global), dtype=torch.float16, device=B.device)
    with torch.cuda.device(B.device):
        err = L.Key_SplitK_API(_stream_ptr(B.device), None, bmp.data_ptr(), NZ.data_ptr(), idx.data_ptr(),
                               NZ_Offset.data_ptr(), B.data_ptr(), C.data_ptr(), M_Global, N, K_Global, None, 1,
                               Batch_Size, num_key_value_groups)
    _lib.check(err, "Key_SplitK_API")
    return C


_workspaces = {}   # (device, stream) -> slab buffer


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """fp32 partial slabs, one buffer per (device, STREAM): launches of one stream run in order, so they may share it;
    two streams (or threads on different streams) never see each other's slabs -- a process-wide buffer did."""
    if torch.cuda.is_current_stream_capturing():   # a capture takes its slabs from its own pool (the graph keeps them for its replays)
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def mustafar_value_formulation(bmp, NZ, idx, NZ_Offset, B, Reduction_Workspace, M_Global: int, K_Global: int,
                               Batch_Size: int, num_key_value_groups: int, split_k: int = 0) -> torch.Tensor:
    """out[b, n, m] = fp16(sum_k Vhat[b // groups][k, m] * B[b, n, k])   (mustafar_wrapper.cu:139-263).

    `split_k` (extension, default 0 = automatic) forces the number of token chunks.
    """
    _common_checks(bmp, NZ, idx, NZ_Offset, B, require_B_contiguous=False)
    if M_Global != 128 or K_Global <= 0 or K_Global % 64:
        raise RuntimeError("mustafar_value_formulation: need M_Global == 128 and K_Global a positive multiple of 64")
    if not B.is_contiguous():
        B = B.contiguous()   # the reference reads the raw pointer as if contiguous
    N = _rows(B, Batch_Size, K_Global)
    _check_cache_shapes(bmp, idx, NZ_Offset, M_Global * K_Global // 64, Batch_Size, num_key_value_groups)
    L = _lib.load()
    if split_k <= 0:
        split_k = L.mustafar_value_pick_split_k(M_Global, N, K_Global, Batch_Size, num_key_value_groups)
    nbytes = L.mustafar_value_workspace_bytes(M_Global, N, K_Global, Batch_Size, num_key_value_groups, split_k)
    ws = _workspace(B.device, nbytes) if nbytes else None
    ws_ptr = ws.data_ptr() if ws is not None else None
    C = torch.empty((Batch_Size, N, M_Global), dtype=torch.float16, device=B.device)
    with torch.cuda.device(B.device):
        err = L.Value_SplitK_API(_stream_ptr(B.device), None, bmp.data_ptr(), NZ.data_ptr(), idx.data_ptr(),
                                 NZ_Offset.data_ptr(), B.data_ptr(), C.data_ptr(), M_Global, N, K_Global, ws_ptr,
                                 split_k, Batch_Size, num_key_value_groups)
    _lib.check(err, "Value_SplitK_API")
    return C
