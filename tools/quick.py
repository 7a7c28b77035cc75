#!/usr/bin/env python3
"""Quick A/B of the fused decode step on the GPU box: one Workload per config, several library settings on it.

    python tools/quick.py --cfg c3 c5 --set valu dot2 mfma valu:lean=0 valu:onepass=0 dot2:tbw=2

A setting is engine[:key=value...] with engine in valu | dot2 | mfma and keys onepass (0 | 1 | 2), lean (0 | 1), tbw, wgs,
winlast, pslab, sb (1: round 5's super-block pair kernel, 0: round 4's pair kernel), small (1: round 6's kernel for launches of two blocks
per workgroup, 0: the super-block kernel there too).
Per setting: self-check against the two reference entry points, then the step replayed as a graph (tokens/s) and the
kernels' own durations.  One line per (config, setting)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mustafar_amd import _lib, mustafar_package as mp  # noqa: E402

ENG = {"valu": 0, "mfma": 1, "dot2": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", nargs="+", default=["c3"])
    ap.add_argument("--set", nargs="+", default=["valu"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--layers", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    timer = bench.KernelTimer(mp)
    timer.install()
    failed = []
    for name in a.cfg:
        w = bench.Workload(name, a.layers, dev, 0, 1, None, False, timer, lib)
        for st in a.set:
            parts = st.split(":")
            kv = dict(p.split("=") for p in parts[1:])
            _lib.check(lib.mustafar_set_fma_engine(ENG[parts[0]]), "engine")
            _lib.check(lib.mustafar_set_onepass(int(kv.get("onepass", 2))), "onepass")
            _lib.check(lib.mustafar_tune(0, int(kv.get("lean", 2))), "lean")
            _lib.check(lib.mustafar_tune(1, int(kv.get("tbw", 0))), "tbw")
            _lib.check(lib.mustafar_tune(2, int(kv.get("wgs", 0))), "wgs")
            _lib.check(lib.mustafar_tune(3, int(kv.get("winlast", 1))), "winlast")
            _lib.check(lib.mustafar_tune(4, int(kv.get("pslab", 0))), "pslab")
            _lib.check(lib.mustafar_tune(8, int(kv.get("sb", 1))), "sb")
            _lib.check(lib.mustafar_tune(9, int(kv.get("late", 1))), "late")
            _lib.check(lib.mustafar_tune(10, int(kv.get("fin1", 1))), "fin1")
            _lib.check(lib.mustafar_tune(11, int(kv.get("small", 1))), "small")
            spec = 0
            if kv.get("spec"):   # (round 6 experiment) spec=1: the measured average bytes of key stream per block; spec=<n>: n bytes
                st0 = w.fused_state()[0]
                spec = int(kv["spec"]) if int(kv["spec"]) > 1 else int(2 * float(st0[0].used.double().mean()) / (st0[4] // 64))
            _lib.check(lib.mustafar_tune(12, spec), "spec")
            ex = w.self_check()
            dt, (ku, vu, n) = w.timed_graph(a.steps, 3)
            rl = w.roofline(ku, vu, n, traffic_file=False)
            print(json.dumps({"cfg": name, "set": st, "self_check_excess": round(ex, 3), "tok_s": round(w.batch * a.steps / dt, 1),
                              "ms_step": round(dt / a.steps * 1e3, 4), "kernel": rl["kernel"], "key_us": round(ku, 2), "value_us": round(vu, 2),
                              "frac": rl["frac"]}), flush=True)
            if not ex <= 1.0:
                failed.append((name, st, ex))
        del w
        torch.cuda.empty_cache()
    if failed:   # (a timing of wrong results is no evidence: the collection scripts stop here)
        raise SystemExit(f"quick.py: self-check FAILED (fused vs the reference entry points, x the fp16 bound): {failed}")


if __name__ == "__main__":
    main()
