#!/usr/bin/env python3
"""Wave timeline of ONE one-pass decode launch (bench.py's c3 layer by default): when every wave starts and ends, how long its
key phase, softmax step and value phase take, how many waves share a SIMD over the span.

    python tools/wave_trace_onepass.py [--cfg c3] [--set valu valu:lean=0 dot2 mfma]
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, "mustafar_amd", "lib", "libmustafar_hip_trace.so")
ENG = {"valu": 0, "mfma": 1, "dot2": 2}


def build_trace_lib():
    srcs = [os.path.join(ROOT, "mustafar_amd", "csrc", f) for f in ("spmv.hip", "compress.hip")]
    if os.path.exists(TRACE_LIB) and all(os.path.getmtime(TRACE_LIB) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                           "-DMUSTAFAR_WAVE_TRACE", "-o", TRACE_LIB] + srcs)


def pct(x, qs=(0, 10, 50, 90, 100)):
    import numpy as np
    return " ".join(f"p{q}={np.percentile(x, q):7.2f}" for q in qs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="c3")
    ap.add_argument("--set", nargs="+", default=["valu"])
    a = ap.parse_args()
    build_trace_lib()
    os.environ["MUSTAFAR_HIP_LIB"] = TRACE_LIB
    import numpy as np
    import torch
    import bench
    from mustafar_amd import _lib, mustafar_package as mp
    lib = _lib.load()
    lib.mustafar_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    timer = bench.KernelTimer(mp)
    w = bench.Workload(a.cfg, 2, dev, 0, 1, None, False, timer, lib)
    state = w.fused_state()
    cap = 1 << 18
    buf = torch.zeros((cap // 4, 16), dtype=torch.int64, device=dev)
    flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)
    for st in a.set:
        parts = st.split(":")
        kv = dict(p.split("=") for p in parts[1:])
        _lib.check(lib.mustafar_set_fma_engine(ENG[parts[0]]), "engine")
        _lib.check(lib.mustafar_set_onepass(int(kv.get("onepass", 1))), "onepass")
        _lib.check(lib.mustafar_tune(0, int(kv.get("lean", 2))), "lean")
        _lib.check(lib.mustafar_tune(1, int(kv.get("tbw", 0))), "tbw")
        _lib.check(lib.mustafar_tune(2, int(kv.get("wgs", 0))), "wgs")
        _lib.check(lib.mustafar_tune(3, int(kv.get("winlast", 1))), "winlast")
        priv = [(p[0], p[1].clone(), p[2], p[3].clone(), p[4], p[5]) for p in state]
        for _ in range(3):
            w.attn.decode_fused(w.qs[0], w.ks[0], w.vs[0], priv[0])
        torch.cuda.synchronize()
        buf.zero_()
        assert lib.mustafar_trace_set(buf.data_ptr(), cap) == 0
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        w.attn.decode_fused(w.qs[1], w.ks[1], w.vs[1], priv[1])
        e1.record()
        torch.cuda.synchronize()
        lib.mustafar_trace_set(None, 0)
        rec = buf.cpu().numpy().astype(np.uint64)
        rec = rec[rec[:, 10] != 0]
        kern = (rec[:, 9] >> np.uint64(56)).astype(int)
        print(f"== {a.cfg} {st}: call {e0.elapsed_time(e1) * 1e3:.1f} us, records {len(rec)} (kernels {sorted(set(kern.tolist()))})")
        base = rec[:, 0].min()
        for k, name in ((7, "lean pair"), (3, "lean"), (4, "pair form"), (6, "matrix-pipe form"), (5, "window workgroups")):
            r = rec[kern == k]
            if not len(r):
                continue
            t = (r[:, :7].astype(np.float64) - float(base)) / 100.0
            has = r[:, 2] != 0   # waves that ran a block
            span = t[:, 6].max()
            life = t[:, 6] - t[:, 0]
            print(f"   {name}: waves={len(r)} (with blocks {has.sum()}) span={span:.2f} us  resident waves/SIMD avg={life.sum() / span / 1024:.2f}")
            print(f"     start      {pct(t[:, 0])}")
            print(f"     end        {pct(t[:, 6])}")
            print(f"     life       {pct(life)}")
            for label, sel in (("first round (start < 5 us)", has & (t[:, 0] < 5.0)), ("later rounds", has & (t[:, 0] >= 5.0))):
                if not sel.any():
                    continue
                th = t[sel]
                print(f"     {label}: {sel.sum()} waves")
                if (r[sel][:, 1] != 0).all():
                    print(f"       -> 1st key chunk staged {pct(th[:, 1] - th[:, 0])}")
                    print(f"       key chunks              {pct(th[:, 2] - th[:, 1])}")
                else:
                    print(f"       key        {pct(th[:, 2] - th[:, 0])}")
                print(f"       softmax    {pct(th[:, 3] - th[:, 2])}")
                if (r[sel][:, 4] != 0).all():
                    print(f"       -> 1st value chunk staged {pct(th[:, 4] - th[:, 3])}")
                    print(f"       value chunks              {pct(th[:, 5] - th[:, 4])}")
                else:
                    print(f"       value      {pct(th[:, 5] - th[:, 3])}")
                print(f"       merge/exit {pct(th[:, 6] - th[:, 5])}")
            # placement: waves per CU and when each CU / XCC is done (HW_ID bits 8..15 = CU, shader array, shader engine; XCC_ID bits 0..3)
            hw = r[:, 8]
            cu = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64) * 256 + ((hw >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
            cus, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
            cu_end = np.array([t[inv == i, 6].max() for i in range(len(cus))])
            print(f"     CUs used: {len(cus)}; waves per CU {pct(cnt)}; last wave of a CU ends {pct(cu_end)}")
            for nw in sorted(set(cnt.tolist())):
                sel_cu = cnt == nw
                print(f"       CUs with {nw:3d} waves: {sel_cu.sum():4d}, end of their last wave {pct(cu_end[sel_cu], (10, 50, 90))}")
            xcc = (cus // 256)
            print("     end of the last wave per XCC: " + " ".join(f"{cu_end[xcc == x].max():6.2f}" for x in sorted(set(xcc.tolist()))))
            edges = np.linspace(0, span, 11)
            occ = [(np.minimum(t[:, 6], edges[i + 1]) - np.maximum(t[:, 0], edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) / 1024 for i in range(10)]
            print("     waves/SIMD per tenth of the span: " + " ".join(f"{o:5.2f}" for o in occ))
            if k == 7:
                # dispatch order: workgroups are handed out by linear id; how do start / end / life depend on it?
                gp = r[:, 9]
                gx = (gp & np.uint64(0xffffff)).astype(np.int64)
                wg = ((gp >> np.uint64(24)) & np.uint64(0xffffff)).astype(np.int64) * (gx.max() + 1) + gx
                nb = 8
                order = np.argsort(wg, kind="stable")
                for name2, col in (("start", t[:, 0]), ("end", t[:, 6]), ("life", life)):
                    eighths = np.array_split(col[order], nb)
                    print(f"     {name2:5s} by workgroup id, eighths (p50): " + " ".join(f"{np.median(x):6.2f}" for x in eighths))
                # age rank of a wave on its SIMD (by start time) against its end
                simd = cu * 4 + ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
                ranks = np.zeros(len(r), dtype=np.int64)
                for s_ in np.unique(simd):
                    ix = np.nonzero(simd == s_)[0]
                    ranks[ix[np.argsort(t[ix, 0], kind="stable")]] = np.arange(len(ix))
                print("     end by age rank on the SIMD (p50): " + " ".join(f"{np.median(t[ranks == q, 6]):6.2f}" for q in range(int(ranks.max()) + 1)))
                print("     life by age rank on the SIMD (p50): " + " ".join(f"{np.median(life[ranks == q]):6.2f}" for q in range(int(ranks.max()) + 1)))
        if os.environ.get("MUSTAFAR_TRACE_DUMP"):
            np.save(os.environ["MUSTAFAR_TRACE_DUMP"] + f"_{a.cfg}_{parts[0]}.npy", rec)


if __name__ == "__main__":
    main()
