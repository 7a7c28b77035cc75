#!/usr/bin/env python3
"""Where a `convert_key_batched` call spends its wall time (round 6): the pieces of the two-pass form timed one by one on the host clock,
the ways of getting the B' + 1 stream offsets to the host, and the whole call in both forms.  c3 shape by default (64 heads x 7936 tokens).

    python tools/probes/convert_breakdown.py [cfg ...]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mustafar_amd import _lib, compression
from tools.microbench import CFG

dev = torch.device("cuda:0")
L = _lib.load()


def wall(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name in sys.argv[1:] or ["c3"]:
    Hq, Hkv, s, Lseq, batch = CFG[name]
    T = ((Lseq - 32) // 256) * 256
    Bp = batch * Hkv
    pr = compression.prune_magnitude(torch.randn((Bp, T, 128), device=dev).half(), s)
    tiles = T * 2
    st = torch.cuda.current_stream().cuda_stream
    res = dict(cfg=name, heads=Bp, tokens=T)
    for which in ("key", "value"):
        fb = getattr(L, f"mustafar_compress_bitmap_{which}")
        fp = getattr(L, f"mustafar_compress_pack_{which}")
        bmp = torch.empty((Bp, tiles), dtype=torch.int64, device=dev)
        acc = torch.empty((Bp, tiles + 1), dtype=torch.int32, device=dev)
        ho = torch.empty((Bp + 1,), dtype=torch.int64, device=dev)
        pin = torch.empty((Bp + 1,), dtype=torch.int64, pin_memory=True)
        stream = torch.cuda.current_stream()

        def alloc3():
            torch.empty((Bp, tiles), dtype=torch.int64, device=dev)
            torch.empty((Bp, tiles + 1), dtype=torch.int32, device=dev)
            torch.empty((Bp + 1,), dtype=torch.int64, device=dev)

        def bitmap():
            fb(st, pr.data_ptr(), Bp, T, 128, bmp.data_ptr(), acc.data_ptr(), ho.data_ptr())

        def bitmap_cpu():
            bitmap()
            return ho.cpu().tolist()

        def bitmap_pin_copy():
            bitmap()
            pin.copy_(ho, non_blocking=True)
            stream.synchronize()
            return pin.tolist()

        def bitmap_into_pinned():      # the kernel's own stores land in host memory
            fb(st, pr.data_ptr(), Bp, T, 128, bmp.data_ptr(), acc.data_ptr(), pin.data_ptr())
            stream.synchronize()
            return pin.tolist()

        offs = bitmap_cpu()
        assert bitmap_pin_copy() == offs and bitmap_into_pinned() == offs
        nz = torch.empty(offs[-1], dtype=torch.float16, device=dev)

        def pack():
            fp(st, pr.data_ptr(), Bp, T, 128, bmp.data_ptr(), acc.data_ptr(), ho.data_ptr(), nz.data_ptr())

        def pack_pinned_offsets():
            fp(st, pr.data_ptr(), Bp, T, 128, bmp.data_ptr(), acc.data_ptr(), pin.data_ptr(), nz.data_ptr())

        ref = nz.clone(); pack(); torch.cuda.synchronize(); ref.copy_(nz)
        nz.zero_(); pack_pinned_offsets(); torch.cuda.synchronize()
        assert torch.equal(ref, nz)
        fn = compression.convert_key_batched if which == "key" else compression.convert_value_batched
        r = {
            "alloc3_host_us": wall(alloc3), "bitmap_kernel_us": wall(bitmap),
            "bitmap+cpu()_us": wall(bitmap_cpu), "bitmap+pinned_copy_us": wall(bitmap_pin_copy), "bitmap_into_pinned_us": wall(bitmap_into_pinned),
            "pack_kernel_us": wall(pack), "pack_offsets_in_host_memory_us": wall(pack_pinned_offsets),
            "pieces_of_host_us": wall(lambda: compression.pieces_of(nz, offs)),
            "call_twopass_us": wall(lambda: compression._convert(pr, which, onepass=False)),
            "call_twopass_sync_by_copy_us": wall(lambda: (setattr(compression, "_CONVERT_SYNC_COPY", True), compression._convert(pr, which, onepass=False),
                                                          setattr(compression, "_CONVERT_SYNC_COPY", False))),
            "call_onepass_us": wall(lambda: compression._convert(pr, which, onepass=True)),
        }
        res[which] = {k: round(v, 1) for k, v in r.items()}
    print(json.dumps(res), flush=True)
