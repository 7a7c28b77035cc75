"""ctypes/numpy front-end of the CPU oracle (oracle/mustafar_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under mustafar_amd/ may import this module.

All arrays are numpy; fp16 data travels as np.float16 (viewed as uint16 for the C side).
Reference citations live next to each C function; the Python layer only marshals.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "mustafar_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.orc_prune_magnitude.argtypes = [vp, vp, i64, i32, i32]
        for name in ("orc_bitmap_key", "orc_bitmap_value"):
            getattr(L, name).argtypes = [vp, i32, i32, i32, vp, vp]
        for name in ("orc_pack_key", "orc_pack_value"):
            getattr(L, name).argtypes = [vp, i32, i32, i32, vp, vp, vp]
        for name in ("orc_key_spmv", "orc_value_spmv"):
            getattr(L, name).argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32]
        L.orc_splitk_reduce.argtypes = [vp, vp, i64, i32, i32]
        L.orc_h2f.argtypes = [ctypes.c_uint16]
        L.orc_h2f.restype = ctypes.c_float
        L.orc_f2h.argtypes = [ctypes.c_float]
        L.orc_f2h.restype = ctypes.c_uint16
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def kth_from_sparsity(target_sparsity: float, D: int) -> int:
    """models/llama_mustafar_kernel.py:97 -- `max(1, int(target_sparsity * D))`."""
    return max(1, int(target_sparsity * D))


def prune_magnitude(x: np.ndarray, target_sparsity: float) -> np.ndarray:
    """dh_prune_key / dh_prune_value (llama_mustafar_kernel.py:77-153) on the last axis."""
    x = _c(x, np.float16)
    D = x.shape[-1]
    out = np.empty_like(x)
    rc = lib().orc_prune_magnitude(_p(x), _p(out), x.size // D, D, kth_from_sparsity(target_sparsity, D))
    if rc:
        raise ValueError(f"orc_prune_magnitude failed rc={rc}")
    return out


def _convert(x: np.ndarray, which: str) -> Tuple[np.ndarray, np.ndarray, List[np.ndarray]]:
    x = _c(x, np.float16)
    assert x.ndim == 3, "inputs must be [B', t, D]"
    B, t, D = x.shape
    assert t % 64 == 0, "M % 64 == 0 (compression.py:253)"
    tiles = t * D // 64
    bmp = np.empty((B, tiles), np.int64)
    accum = np.empty((B, tiles + 1), np.int32)
    rc = getattr(lib(), f"orc_bitmap_{which}")(_p(x), B, t, D, _p(bmp), _p(accum))
    if rc:
        raise ValueError(f"orc_bitmap_{which} failed rc={rc}")
    lens = 2 * accum[:, -1].astype(np.int64)
    flat = np.empty(int(lens.sum()), np.float16)
    rc = getattr(lib(), f"orc_pack_{which}")(_p(x), B, t, D, _p(bmp), _p(accum), _p(flat))
    if rc:
        raise ValueError(f"orc_pack_{which} failed rc={rc}")
    offs = np.concatenate([[0], np.cumsum(lens)])
    nzs = [flat[offs[b]:offs[b + 1]].copy() for b in range(B)]
    return bmp, accum, nzs


def convert_key_batched(x: np.ndarray):
    """kernel/compression.py:249-339 -> (bitmaps i64 [B',2t], accum_counts i32 [B',2t+1], list of fp16)."""
    return _convert(x, "key")


def convert_value_batched(x: np.ndarray):
    """kernel/compression.py:341-432."""
    return _convert(x, "value")


def nz_offset_from_idx(accum: np.ndarray) -> np.ndarray:
    """llama_mustafar_kernel.py:329-331 / :423-425: nz_offset[i] = nz_offset[i-1] + idx[i-1][-1] // 4."""
    last = accum.reshape(accum.shape[0], -1)[:, -1].astype(np.int64) // 4
    out = np.zeros(accum.shape[0], np.int32)
    out[1:] = np.cumsum(last)[:-1]
    return out


def _spmv(which, bmp, nz_flat, idx, nz_offset, Bm, M_Global, K_Global, Batch_Size, groups):
    bmp = _c(bmp, np.int64).reshape(-1)
    idx = _c(idx, np.int32).reshape(-1)
    nz_flat = _c(nz_flat, np.float16).reshape(-1)
    nz_offset = _c(nz_offset, np.int32)
    Bm = _c(Bm, np.float16)
    N = Bm.shape[-2]
    C = np.zeros((Batch_Size, N, M_Global), np.float16)
    Cd = np.zeros((Batch_Size, N, M_Global), np.float64)
    rc = getattr(lib(), f"orc_{which}_spmv")(_p(bmp), _p(nz_flat), _p(idx), _p(nz_offset), _p(Bm), _p(C), _p(Cd),
                                             M_Global, N, K_Global, Batch_Size, groups)
    if rc:
        raise ValueError(f"orc_{which}_spmv failed rc={rc}")
    return C, Cd


def key_spmv(bmp, nz_flat, idx, nz_offset, Bm, M_Global, K_Global, Batch_Size, groups):
    """mustafar_key_formulation semantics; returns (C fp16 [Batch,N,M], same sums in float64)."""
    return _spmv("key", bmp, nz_flat, idx, nz_offset, Bm, M_Global, K_Global, Batch_Size, groups)


def value_spmv(bmp, nz_flat, idx, nz_offset, Bm, M_Global, K_Global, Batch_Size, groups):
    """mustafar_value_formulation semantics; returns (C fp16 [Batch,N,M], float64 sums)."""
    return _spmv("value", bmp, nz_flat, idx, nz_offset, Bm, M_Global, K_Global, Batch_Size, groups)


# ---------------------------------------------------------------- independent numpy decompressors
# Written separately from the C code (vectorised bit unpack) so tests can cross-check the two.

def _bits(bmp_row: np.ndarray) -> np.ndarray:
    """[tiles] int64 -> [tiles, 64] bool, element i <-> bit 63-i."""
    u = bmp_row.astype(np.uint64)
    sh = np.arange(63, -1, -1, dtype=np.uint64)
    return ((u[:, None] >> sh[None, :]) & np.uint64(1)).astype(bool)


def _tiles_dense(bmp_row, accum_row, nz) -> np.ndarray:
    bits = _bits(bmp_row)
    tiles = bits.shape[0]
    dense = np.zeros((tiles, 64), np.float16)
    rank = np.cumsum(bits, axis=1) - 1
    src = 2 * accum_row[:-1].astype(np.int64)[:, None] + rank
    dense[bits] = nz[src[bits]]
    return dense


def decompress_key(bmp, accum, nzs, t: int, D: int) -> np.ndarray:
    """-> pruned K [B', t, D]; tile id = tokblk*D + d (compression.py:32-36)."""
    B = len(nzs)
    bmp = np.asarray(bmp).reshape(B, -1)
    accum = np.asarray(accum).reshape(B, -1)
    out = np.zeros((B, t, D), np.float16)
    for b in range(B):
        dense = _tiles_dense(bmp[b], accum[b], np.asarray(nzs[b]))          # [tokblk*D + d, 64 tokens]
        out[b] = dense.reshape(t // 64, D, 64).transpose(0, 2, 1).reshape(t, D)
    return out


def decompress_value(bmp, accum, nzs, t: int, D: int) -> np.ndarray:
    """-> pruned V [B', t, D]; tile id = tokblk*(D/64)*64 + half*64 + r (compression.py:87-97)."""
    B = len(nzs)
    bmp = np.asarray(bmp).reshape(B, -1)
    accum = np.asarray(accum).reshape(B, -1)
    out = np.zeros((B, t, D), np.float16)
    for b in range(B):
        dense = _tiles_dense(bmp[b], accum[b], np.asarray(nzs[b]))          # [tokblk, half, r, 64 ch]
        out[b] = dense.reshape(t // 64, D // 64, 64, 64).transpose(0, 2, 1, 3).reshape(t, D)
    return out
