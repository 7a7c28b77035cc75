#!/usr/bin/env python3
"""Throughput of the prefill-side kernels (prune, bitmap+scan, pack) at BASELINE sizes; HBM bytes = in + out."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mustafar_amd import _lib, compression
from tools.microbench import CFG, timeit

dev = torch.device("cuda:0")
L = _lib.load()
for name in sys.argv[1:] or ["c3", "c4"]:
    Hq, Hkv, s, Lseq, batch = CFG[name]
    T = ((Lseq - 32) // 256) * 256
    Bp = batch * Hkv
    x = torch.randn((Bp, T, 128), device=dev).half()
    out = torch.empty_like(x)
    t_prune = timeit(lambda: compression.prune_magnitude(x, s, out=out), 10)
    pr = compression.prune_magnitude(x, s)
    tiles = T * 2
    st = torch.cuda.current_stream().cuda_stream
    res = dict(cfg=name, rows=Bp * T, prune_us=round(t_prune * 1e6, 1), prune_GBps=round(2 * x.numel() * 2 / t_prune / 1e9, 1))
    for which in () if os.environ.get("FUSED_ONLY") else ("key", "value"):
        bmp = torch.empty((Bp, tiles), dtype=torch.int64, device=dev)
        acc = torch.empty((Bp, tiles + 1), dtype=torch.int32, device=dev)
        ho = torch.empty((Bp + 1,), dtype=torch.int64, device=dev)
        fb = getattr(L, f"mustafar_compress_bitmap_{which}")
        fp = getattr(L, f"mustafar_compress_pack_{which}")
        t_b = timeit(lambda: fb(st, pr.data_ptr(), Bp, T, 128, bmp.data_ptr(), acc.data_ptr(), ho.data_ptr()), 10)
        total = int(ho.cpu()[-1])
        nz = torch.empty(total, dtype=torch.float16, device=dev)
        t_p = timeit(lambda: fp(st, pr.data_ptr(), Bp, T, 128, bmp.data_ptr(), acc.data_ptr(), ho.data_ptr(), nz.data_ptr()), 10)
        t_full = timeit(lambda: (compression.convert_key_batched if which == "key" else compression.convert_value_batched)(pr), 5)
        inb = pr.numel() * 2
        res.update({f"{which}_bitmap_scan_us": round(t_b * 1e6, 1), f"{which}_bitmap_GBps": round((inb + bmp.numel() * 8 + acc.numel() * 4) / t_b / 1e9, 1),
                    f"{which}_pack_us": round(t_p * 1e6, 1), f"{which}_pack_GBps": round((inb + acc.numel() * 4 + total * 2) / t_p / 1e9, 1),
                    f"{which}_convert_call_us": round(t_full * 1e6, 1)})
    # fused form: raw rows -> prune thresholds in registers -> bitmaps/offsets -> pack, K and V in the same three launches
    from mustafar_amd.cache import CompressedArena
    xk = torch.randn((batch, Hkv, T, 128), device=dev).half()
    xv = torch.randn((batch, Hkv, T, 128), device=dev).half()
    kth = compression.kth_from_sparsity(s, 128)
    ka, va = CompressedArena.from_raw_pair(xk, xv, T, kth, kth)       # sizes the regions
    scratch = torch.empty(int(L.mustafar_compress_scratch_bytes(Bp, T)), dtype=torch.uint8, device=dev)

    def fused():
        _lib.check(L.mustafar_cache_append_kv(st, xk.data_ptr(), xv.data_ptr(), T * 128, Bp, T, 128, kth, kth, ka.view_ptr(), va.view_ptr(), 0,
                                              ka._totals.data_ptr(), va._totals.data_ptr(), ka.nz_cap, va.nz_cap, ka._overflow.data_ptr(), scratch.data_ptr()), "append_kv")
    t_f = timeit(fused, 10)
    assert int(ka._overflow) == 0
    # bytes of the one-pass form: the raw K and V blocks are read ONCE each; bitmaps (8 B), offsets (4 B) per tile and the packed
    # streams are written once
    in_b = (xk.numel() + xv.numel()) * 2
    out_b = int(ka.used.sum() + va.used.sum()) * 2 + 2 * Bp * tiles * 12
    res.update(fused_prune_compress_kv_us=round(t_f * 1e6, 1), fused_per_side_us=round(t_f * 1e6 / 2, 1),
               fused_bytes_in=in_b, fused_bytes_out=out_b, fused_GBps_in_once_plus_out=round((in_b + out_b) / t_f / 1e9, 1),
               fused_frac_of_8TBps=round((in_b + out_b) / t_f / 8e12, 3))
    print(json.dumps(res), flush=True)
