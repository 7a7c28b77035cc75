#!/usr/bin/env python3
"""Burst vs sustained: what clock do the one-pass launches run at, 25 ms into a run and 0.5 s into it?  (round 6)

The bench's 20-step headline is a ~27-ms burst behind a synchronisation point; from ~100 steps on the same step reads 5-8 % slower
(profiles/r05_probes.txt item 21).  rocm-smi's sclk does not answer why (MI355X_MICROARCH.md, DVFS give-back: the in-kernel clock reads up to
~10 % below pp_dpm_sclk).  This tool measures the clock INSIDE the kernel: the diagnostic library (-DMUSTAFAR_WAVE_TRACE, never the product
library) stamps s_memtime (shader clock) next to s_memrealtime (100 MHz) at the start and the end of every wave of the one-pass launch; a
wave's clock is d(s_memtime) / d(s_memrealtime) x 100 MHz.  The trace buffer holds the LAST launch of a run (every launch overwrites it), so a
run of n replays of the c3 step (32 launches each, fixed compressed length and window: bench.Workload.fixed_graph) behind an idle pause gives
the clock n steps into a run.

    python tools/clock_probe.py [--cfg c3] [--runs 5 20 60 100 200 400] [--idle 0.5]

Also printed: ms/step of every run with the diagnostic library and -- the numbers that matter -- with the PRODUCT library in a child process
(--product: the course of ms/step over a 400-step run, HIP events every 20 replays)."""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def product_course(cfg, steps, idle):
    import torch
    import bench
    from mustafar_amd import _lib, mustafar_package as mp
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    w = bench.Workload(cfg, 32, dev, 0, 1, None, False, bench.KernelTimer(mp), lib)
    g, _, state = w.fixed_graph()
    for _ in range(5):
        g.replay()
    out = {}
    for n in (20, steps):
        torch.cuda.synchronize()
        time.sleep(idle)
        per = 20
        evs = []
        t0 = time.perf_counter()
        for i in range(n):
            if i % per == 0:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
            g.replay()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        out[str(n)] = {"ms_per_step_wall": round(wall / n * 1e3, 4), "ms_per_step_by_20": [round(evs[i].elapsed_time(evs[i + 1]) / per, 4) for i in range(len(evs) - 1)]}
    print(json.dumps({"library": "product", "cfg": cfg, "idle_s": idle, "runs": out}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="c3")
    ap.add_argument("--runs", type=int, nargs="+", default=[5, 20, 60, 100, 200, 400])
    ap.add_argument("--idle", type=float, default=0.5)
    ap.add_argument("--product", action="store_true")
    a = ap.parse_args()
    if a.product:
        return product_course(a.cfg, max(a.runs), a.idle)
    import wave_trace_onepass as wt
    wt.build_trace_lib()
    # the product library's own course first, in a child process of its own (one library per process)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--product", "--cfg", a.cfg, "--runs", str(max(a.runs)), "--idle", str(a.idle)])
    os.environ["MUSTAFAR_HIP_LIB"] = wt.TRACE_LIB
    import numpy as np
    import torch
    import bench
    from mustafar_amd import _lib, mustafar_package as mp
    lib = _lib.load()
    lib.mustafar_trace_set.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    w = bench.Workload(a.cfg, 32, dev, 0, 1, None, False, bench.KernelTimer(mp), lib)
    g, _, state = w.fixed_graph()
    cap = 1 << 18
    buf = torch.zeros((cap // 4, 16), dtype=torch.int64, device=dev)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    for rep in range(2):
        for n in a.runs:
            torch.cuda.synchronize()
            time.sleep(a.idle)
            buf.zero_()
            torch.cuda.synchronize()
            assert lib.mustafar_trace_set(buf.data_ptr(), cap) == 0
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            lib.mustafar_trace_set(None, 0)
            rec = buf.cpu().numpy().astype(np.uint64)
            rec = rec[rec[:, 10] != 0]
            kern = (rec[:, 9] >> np.uint64(56)).astype(int)
            r = rec[(kern == 7) & (rec[:, 2] != 0)]          # SpMV waves of the one-pass launch that ran a block
            d_real = (r[:, 6] - r[:, 0]).astype(np.float64)  # 100 MHz ticks
            d_clk = (r[:, 12] - r[:, 11]).astype(np.float64)
            ok = d_real >= 500                                # (waves that lived >= 5 us: a tick is 10 ns)
            mhz = d_clk[ok] / d_real[ok] * 100.0
            span = (r[:, 6].max() - r[:, 0].min()) / 100.0
            print(json.dumps({"library": "trace", "rep": rep, "replays": n, "ms_per_step_wall": round(wall / n * 1e3, 4), "at_ms": round(wall * 1e3, 1),
                              "last_launch_span_us": round(float(span), 2), "waves": int(ok.sum()),
                              "in_kernel_clock_MHz": {"p10": round(float(np.percentile(mhz, 10)), 1), "p50": round(float(np.percentile(mhz, 50)), 1),
                                                      "p90": round(float(np.percentile(mhz, 90)), 1)}}), flush=True)


if __name__ == "__main__":
    main()
