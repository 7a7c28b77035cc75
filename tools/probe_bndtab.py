#!/usr/bin/env python3
"""Probe (round 4b): what would a compact, L2-hot table of chunk bounds per block be worth to the one-pass launch?

A wave's first vector load is its block's chunk bounds -- five strided reads of the offsets array, HBM-cold -- and its first
stream load depends on them: two dependent round trips before the first FMA.  The probe library (tools/build_variant.sh bndtab
-DMUSTAFAR_PROBE_BNDTAB, with the patch tools/probes/bndtab.patch applied to spmv.hip) reads the bounds of a block from a table
[kv-head][block][8] (32 bytes per block) registered per cache; this tool builds the tables from the caches' own offsets and times
the same workload without and with them, in one process.

    MUSTAFAR_HIP_LIB=mustafar_amd/lib/variants/libmustafar_hip_bndtab.so python tools/probe_bndtab.py --cfg c3 c2
"""
import argparse
import ctypes
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
    ap.add_argument("--set", nargs="+", default=["dot2", "mfma"])
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    lib.mustafar_probe_bndtab.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.mustafar_probe_bndtab.restype = ctypes.c_int
    timer = bench.KernelTimer(mp)
    timer.install()
    for name in a.cfg:
        w = bench.Workload(name, 32, dev, 0, 1, None, False, timer, lib)
        state = w.fused_state()

        def same_arenas():   # every leg on the SAME arenas (the tables are registered by the address of a cache's offsets), fresh windows
            for q in state:
                for arena in (q[0], q[2]):
                    arena.drop_extents()
            w.cfg.api, w.cfg.arena = "fused", True
            return [w.attn.to_fused((q[0], o[1], q[2], o[3], q[4], q[5])) for q, o in zip(state, w.pasts)]
        w.fused_state = same_arenas
        tables = []
        for p in state:
            for arena in (p[0], p[2]):
                ntb = arena.tokens // 64
                cols = (torch.arange(ntb, device=dev)[:, None] * 128 + torch.arange(5, device=dev)[None, :] * 32).reshape(-1)
                t = torch.zeros((arena.idx.shape[0], ntb, 8), dtype=torch.int32, device=dev)
                t[:, :, :5] = arena.idx[:, cols].view(arena.idx.shape[0], ntb, 5)
                tables.append((arena.idx.data_ptr(), t))
        for st in a.set:
            _lib.check(lib.mustafar_set_fma_engine(ENG[st]), "engine")
            for mode in ("offsets", "table"):
                lib.mustafar_probe_bndtab(None, None)
                if mode == "table":
                    for ptr, t in tables:
                        _lib.check(lib.mustafar_probe_bndtab(ptr, t.data_ptr()), "bndtab")
                ex = w.self_check()
                if mode == "table":   # (a validation build -- MUSTAFAR_PROBE_BNDTAB=2 -- flags a table entry that differs from the offsets in slot 7)
                    bad = [(i, int(t[:, :, 7].count_nonzero())) for i, (_, t) in enumerate(tables) if bool(t[:, :, 7].any())]
                    if bad:
                        i, _ = bad[0]
                        h, tb = [int(x[0]) for x in torch.nonzero(tables[i][1][:, :, 7], as_tuple=True)]
                        print(json.dumps({"cfg": name, "mismatching_tables": bad[:8], "first": [i, h, tb], "entry": tables[i][1][h, tb].tolist()}), flush=True)
                        return
                dt, (ku, vu, n) = w.timed_graph(a.steps, 3)
                if mode == "table":
                    bad = [(i, int(t[:, :, 7].count_nonzero())) for i, (_, t) in enumerate(tables) if bool(t[:, :, 7].any())]
                    if bad:
                        i, _ = bad[0]
                        nzs = torch.nonzero(tables[i][1][:, :, 7])
                        h, tb = int(nzs[0][0]), int(nzs[0][1])
                        print(json.dumps({"cfg": name, "after": "timed_graph", "mismatching_tables": bad[:8], "first": [i, h, tb], "n_in_first": int(nzs.shape[0]),
                                          "entry": tables[i][1][h, tb].tolist(), "last": [int(nzs[-1][0]), int(nzs[-1][1])]}), flush=True)
                        return
                print(json.dumps({"cfg": name, "set": st, "bounds": mode, "self_check_excess": round(ex, 3), "tok_s": round(w.batch * a.steps / dt, 1),
                                  "kernel_us": round(ku, 2)}), flush=True)
        lib.mustafar_probe_bndtab(None, None)
        del w, state, tables
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
