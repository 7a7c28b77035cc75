"""ctypes binding of libmustafar_hip.so (the C ABI declared in include/mustafar_hip.h).

There is no CPU fallback: if the HIP library is missing or a symbol is absent this module raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MUSTAFAR_HIP_LIB points at another build of the same ABI (e.g. the wave-trace build of tools/wave_trace.py)
LIB_PATH = os.environ.get("MUSTAFAR_HIP_LIB") or os.path.join(_HERE, "lib", "libmustafar_hip.so")

_vp, _i32, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64


class CacheView(ctypes.Structure):
    """`mustafar_cache_view` of include/mustafar_hip.h: the four arrays of the compressed format + the head strides."""
    _fields_ = [("bmp", _vp), ("nz", _vp), ("idx", _vp), ("nz_offset", _vp),
                ("bmp_head_stride", _i64), ("idx_head_stride", _i64), ("nz_head_stride", _i64)]


_view_p = ctypes.POINTER(CacheView)


class TriggerItem(ctypes.Structure):
    """`mustafar_trigger_item` of include/mustafar_hip.h: one layer of a batched 256-token trigger."""
    _fields_ = [("k_window", _vp), ("v_window", _vp), ("k_dst", CacheView), ("v_dst", CacheView), ("k_table_slot", _vp), ("v_table_slot", _vp),
                ("k_head_total", _vp), ("v_head_total", _vp), ("overflow_flag", _vp)]


_item_p = ctypes.POINTER(TriggerItem)

# name -> (restype, argtypes); must list every symbol of include/mustafar_hip.h
SIGNATURES = {
    "mustafar_abi_version": (_i32, []),
    "Key_SplitK_API": (_i32, [_vp] * 8 + [_i32] * 3 + [_vp] + [_i32] * 3),
    "Value_SplitK_API": (_i32, [_vp] * 8 + [_i32] * 3 + [_vp] + [_i32] * 3),
    "mustafar_value_pick_split_k": (_i32, [_i32] * 5),
    "mustafar_value_workspace_bytes": (_i64, [_i32] * 6),
    "mustafar_decode_attention": (_i32, [_vp] * 14 + [_i32, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, ctypes.c_float, _vp,
                                         _vp, _i64, _i32, ctypes.c_uint32]),
    "mustafar_decode_attention_view": (_i32, [_vp, _view_p, _view_p] + [_vp] * 5 + [_i32, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32,
                                              ctypes.c_float, _vp, _vp, _i64, _i32, ctypes.c_uint32]),
    "mustafar_decode_attention_extents": (_i32, [_vp, _view_p, _view_p, _i32, _vp, _vp] + [_vp] * 5 +
                                          [_i32, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, ctypes.c_float, _vp, _vp, _i64, _i32,
                                           ctypes.c_uint32, _vp]),
    "mustafar_decode_reads_extents": (_i32, [_i32, _i32, ctypes.c_uint32]),
    "mustafar_cache_append_bitmap_key": (_i32, [_vp, _vp, _i32, _i32, _i32, _view_p, _i32, _vp]),
    "mustafar_cache_append_bitmap_value": (_i32, [_vp, _vp, _i32, _i32, _i32, _view_p, _i32, _vp]),
    "mustafar_cache_append_pack_key": (_i32, [_vp, _vp, _i32, _i32, _i32, _view_p, _i32]),
    "mustafar_cache_append_pack_value": (_i32, [_vp, _vp, _i32, _i32, _i32, _view_p, _i32]),
    "mustafar_counter_add": (_i32, [_vp, _vp, _i32]),
    "mustafar_compress_scratch_bytes": (_i64, [_i32, _i32]),
    "mustafar_cache_append_kv": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _view_p, _view_p, _i32, _vp, _vp, _i64, _i64, _vp, _vp]),
    "mustafar_window_drop_front": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32]),
    "mustafar_trigger_compress_batch": (_i32, [_vp, _i32, _item_p, _i64, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _vp]),
    "mustafar_trigger_finish_batch": (_i32, [_vp, _i32, _item_p, _i64, _i32, _i32, _i32]),
    "mustafar_cache_rehouse": (_i32, [_vp, _view_p, _view_p, _i32, _i32, _i64]),
    "mustafar_decode_workspace_bytes": (_i64, [_i32] * 4),
    "mustafar_set_fma_engine": (_i32, [_i32]),
    "mustafar_get_fma_engine": (_i32, []),
    "mustafar_set_onepass": (_i32, [_i32]),
    "mustafar_get_onepass": (_i32, []),
    "mustafar_last_decode_choice": (_i32, []),
    "mustafar_tune": (_i32, [_i32, _i32]),
    "mustafar_profile_begin": (_i32, [_i32]),
    "mustafar_profile_end": (_i32, [_vp, _vp, _vp]),
    "mustafar_profile_end2": (_i32, [_vp, _vp, _vp, _vp]),
    "mustafar_prune_magnitude": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32]),
    "mustafar_compress_bitmap_key": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mustafar_compress_bitmap_value": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mustafar_compress_bitmap_mirrored": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mustafar_compress_pack_key": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mustafar_compress_pack_value": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mustafar_cache_consolidate_extents": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i64]),
    "mustafar_compress_set_form": (_i32, [_i32]),
    "mustafar_compress_get_form": (_i32, []),
    "mustafar_compress_test_skip_publish": (_i32, [_i32]),
    "mustafar_convert_scratch_bytes": (_i64, [_i32, _i32]),
    "mustafar_convert_onepass": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mustafar_convert_onepass_mirrored": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mustafar_convert_pack": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
}

# `flags` of the fused entry points (MUSTAFAR_FLAG_* in include/mustafar_hip.h)
ENGINE_FLAGS = {None: 0, "default": 0, "valu": 1, "fma_mix": 1, "mfma": 2, "dot2": 3}
STRUCTURE_FLAGS = {None: 0, "auto": 0, "two_launch": 1 << 4, "one_pass": 2 << 4}

_lib = None


class MustafarLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises MustafarLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MustafarLibraryError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (or mustafar_amd/csrc/build.sh). "
            "mustafar_amd has no CPU fallback.")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. ROCm runtime libraries missing
        raise MustafarLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise MustafarLibraryError(f"{LIB_PATH} lacks symbol {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(err: int, what: str) -> None:
    if err != 0:
        raise RuntimeError(f"{what} failed: HIP error {err}" + (" (invalid argument)" if err == 1 else ""))
