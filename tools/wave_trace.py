#!/usr/bin/env python3
"""Wave timeline of the two SpMV kernels: when every wave starts and ends, and where it ran.

Builds (once) a -DMUSTAFAR_WAVE_TRACE copy of the library next to the normal one, runs ONE key and ONE value SpMV
of a BASELINE config against an HBM-cold cache and prints start / end / duration percentiles, the average number of
resident waves per SIMD, and the spread over XCDs and CUs.  This answers "ramp, tail or imbalance?" when a kernel is
slower than its instruction count says (rocprofv3 only gives the launch duration).

    python tools/wave_trace.py [--cfg c5] [--split 0] [--dump gpurun_out/trace.npz]
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, "mustafar_amd", "lib", "libmustafar_hip_trace.so")


def build_trace_lib():
    srcs = [os.path.join(ROOT, "mustafar_amd", "csrc", f) for f in ("spmv.hip", "compress.hip")]
    if os.path.exists(TRACE_LIB) and all(os.path.getmtime(TRACE_LIB) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                           "-DMUSTAFAR_WAVE_TRACE", "-o", TRACE_LIB] + srcs)


def pct(x, qs=(0, 10, 50, 90, 100)):
    import numpy as np
    return " ".join(f"p{q}={np.percentile(x, q):8.2f}" for q in qs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="c5")
    ap.add_argument("--split", type=int, default=0, help="Split_K of the value SpMV (0 = the library's choice)")
    ap.add_argument("--dump", default=None)
    args = ap.parse_args()

    build_trace_lib()
    os.environ["MUSTAFAR_HIP_LIB"] = TRACE_LIB
    import numpy as np
    import torch
    from mustafar_amd import _lib, mustafar_package as mp
    from tools.microbench import CFG, build_cache

    L = _lib.load()
    L.mustafar_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1)
    Hq, Hkv, s, Lseq, batch = CFG[args.cfg]
    T = ((Lseq - 32) // 256) * 256
    Bp, BH, groups = batch * Hkv, batch * Hq, Hq // Hkv
    kc = build_cache(Bp, T, s, "key", dev, gen)
    vc = build_cache(Bp, T, s, "value", dev, gen)
    flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)   # 1 GiB: evicts L2 and the Infinity Cache
    ws = torch.zeros(1, dtype=torch.float16, device=dev)
    q = torch.randn((BH, 1, 128), device=dev, generator=gen).half()
    p = torch.softmax(torch.randn((BH, 1, T), device=dev, generator=gen), -1).half()
    cap = 1 << 19
    buf = torch.zeros((cap, 4), dtype=torch.int64, device=dev)
    split = args.split or L.mustafar_value_pick_split_k(128, 1, T, BH, groups)

    def run():
        mp.mustafar_key_formulation(*kc, q, T, 128, BH, groups)
        mp.mustafar_value_formulation(*vc, p, ws, 128, T, BH, groups, split_k=split)

    run()   # warm-up (code objects, occupancy query)
    torch.cuda.synchronize()
    assert L.mustafar_trace_set(buf.data_ptr(), cap) == 0
    iters = 5   # the first launches after an idle gap run at low clocks; the last launch overwrites the earlier ones
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(iters):
        flush.zero_()
        if it == iters - 1:
            evs[0].record()
            mp.mustafar_key_formulation(*kc, q, T, 128, BH, groups)
            evs[1].record()
            mp.mustafar_value_formulation(*vc, p, ws, 128, T, BH, groups, split_k=split)
            evs[2].record()
        else:
            run()
    torch.cuda.synchronize()
    print(f"event timing of the traced launches: key {evs[0].elapsed_time(evs[1]) * 1e3:.1f} us, "
          f"value (+combine) {evs[1].elapsed_time(evs[2]) * 1e3:.1f} us")
    rec = buf.cpu().numpy().astype(np.uint64)
    rec = rec[rec[:, 1] != 0]   # slots are grid positions; a later launch overwrites an earlier one
    L.mustafar_trace_set(None, 0)
    print(f"cfg={args.cfg} T={T} B'={Bp} BH={BH} value Split_K={split} records={len(rec)}")
    if args.dump:
        np.savez_compressed(args.dump, rec=rec)

    kern = (rec[:, 3] >> np.uint64(56)).astype(int)
    for k, name in ((1, "key_spmv_kernel"), (2, "value_spmv_kernel")):
        r = rec[kern == k]
        if not len(r):
            continue
        t0 = (r[:, 0] - r[:, 0].min()).astype(np.float64) / 100.0    # 100 MHz -> us
        t1 = (r[:, 1] - r[:, 0].min()).astype(np.float64) / 100.0
        dur = t1 - t0
        hw = r[:, 2] & np.uint64(0xFFFFFFFF)
        xcc = ((r[:, 2] >> np.uint64(32)) & np.uint64(0xF)).astype(int)
        cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int)
        sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(int)
        se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(int)
        simd = ((hw >> np.uint64(4)) & np.uint64(0x3)).astype(int)
        span = t1.max()
        print(f"== {name}: waves={len(r)} span={span:.2f} us  resident waves/SIMD avg={dur.sum() / span / 1024:.2f}")
        print(f"   start us   {pct(t0)}")
        print(f"   end us     {pct(t1)}")
        print(f"   life us    {pct(dur)}")
        # occupancy over time: resident waves per SIMD in 10 slices of the span
        edges = np.linspace(0, span, 11)
        occ = [(np.minimum(t1, edges[i + 1]) - np.maximum(t0, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) / 1024
               for i in range(10)]
        print("   waves/SIMD per tenth of the span: " + " ".join(f"{o:5.2f}" for o in occ))
        cu_id = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        ids, cnt = np.unique(cu_id, return_counts=True)
        last = np.array([t1[cu_id == i].max() for i in ids])
        print(f"   CUs seen={len(ids)} waves per CU min={cnt.min()} max={cnt.max()}  last end per CU {pct(last)}")
        for x in range(8):
            m = xcc == x
            if m.any():
                print(f"   xcc{x}: waves={m.sum():6d} first start={t0[m].min():7.2f} last end={t1[m].max():7.2f} life p50={np.median(dur[m]):7.2f}")
        _ = simd


if __name__ == "__main__":
    main()
