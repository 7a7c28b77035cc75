#!/usr/bin/env python3
"""Instruction classes of a kernel's ISA, per basic block and over the path one wave takes through the block loop.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o spmv.s mustafar_amd/csrc/spmv.hip
    python tools/isa_breakdown.py spmv.s 'decode_onepass_leanpair_kernelILi2ELb0ELi4E' [--path even|odd]

The block loop of the pair form is ONE loop (the only depth-1 loop that holds s_barrier); inside it the compiler lays the even wave's
and the odd wave's key / value phases out as alternatives behind wave-uniform branches.  A wave's dynamic path = the loop's blocks
minus the other parity's phase bodies (the two largest alternatives of each pair), minus the mask branch.  Per-tile figures divide by
the 128 tiles (64 key + 64 value) a wave works through per block.
"""
import argparse
import collections
import re
import sys

CLASSES = [
    ("v_mbcnt", r"v_mbcnt_"),
    ("v_lshl_add_u32 (gather address)", r"v_lshl_add_u32"),
    ("v_mov_b32 .., 0 (zero)", r"v_mov_b32(_e32)? v\d+, 0$"),
    ("v_or_b32", r"v_or_b32"),
    ("v_dot2_f32_f16", r"v_dot2_f32_f16"),
    ("v_fma_mix", r"v_fma_mix"),
    ("v_mfma", r"v_mfma"),
    ("v_readlane (scalar spill reload / bounds)", r"v_readlane_b32"),
    ("v_writelane (scalar spill)", r"v_writelane_b32"),
    ("v_readfirstlane", r"v_readfirstlane"),
    ("DPP (cross-lane reductions)", r"_dpp|row_shr|row_bcast|quad_perm"),
    ("v_exp / v_rcp / v_log", r"v_exp_|v_rcp_|v_log_"),
    ("v_cvt", r"v_cvt_"),
    ("v_cndmask", r"v_cndmask"),
    ("v_lshl_add_u64 / v_add (addresses)", r"v_lshl_add_u64|v_add_co|v_addc_co|v_add_u32|v_ashrrev|v_lshlrev"),
    ("other VALU", r"v_"),
]
MEMC = [
    ("ds_read_u16 (gather)", r"ds_read_u16"),
    ("ds_read other", r"ds_read"),
    ("ds_write", r"ds_write"),
    ("buffer_load", r"buffer_load"),
    ("global_load", r"global_load"),
    ("global_store", r"global_store"),
    ("s_load", r"s_load"),
    ("s_waitcnt", r"s_waitcnt"),
    ("s_barrier", r"s_barrier"),
    ("s_nop", r"s_nop"),
    ("s_mov_b64 exec", r"s_mov_b64 exec"),
    ("s_brev_b64", r"s_brev_b64"),
    ("other SALU", r"s_"),
]


def classify(ins):
    if ins.startswith("v_"):
        for name, pat in CLASSES:
            if re.search(pat, ins):
                return "V", name
    for name, pat in MEMC:
        if re.match(pat, ins) or (pat.startswith("s_mov_b64 exec") and ins.startswith("s_mov_b64 exec")):
            return "M", name
    return "M", "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--tiles", type=int, default=128, help="tiles one wave works through per trip of the block loop")
    ap.add_argument("--one-body", action="store_true", help="the loop holds ONE body for both waves of a pair (decode_onepass_sb_kernel): every "
                    "block of the loop is on the path; without it the second and fourth largest blocks (the other parity's phases) are left out")
    ap.add_argument("--markers", action="store_true", help="count what lies between the `; sb_trips_begin` and `; sb_trips_end` comments of the "
                    "kernel's text (decode_onepass_sb_kernel: with one trip per pair there is no loop to find); implies --one-body")
    a = ap.parse_args()
    lines = open(a.asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(a.kernel) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start + 1:end]
    if a.markers:
        b0 = next(i for i, l in enumerate(body) if "sb_trips_begin" in l)
        b1 = next(i for i, l in enumerate(body) if "sb_trips_end" in l)
        tot = collections.Counter()
        for l in body[b0:b1]:
            t = l.strip()
            if not t or t.startswith(";") or t.startswith(".") or re.match(r"^\.?LBB", t):
                continue
            t = t.split(";")[0].strip()
            if t:
                tot[classify(t)] += 1
        v = sum(k for (c, _), k in tot.items() if c == "V")
        print(f"kernel {a.kernel}: between the trip markers {sum(tot.values())} instructions, {v} vector = {v / a.tiles:.2f} per tile ({a.tiles} tiles per wave and trip)")
        for (c, name), k in sorted(tot.items(), key=lambda x: (x[0][0] != "V", -x[1])):
            print(f"   {'VALU' if c == 'V' else '    '}  {name:48s} {k:5d}   {k / a.tiles:6.3f} per tile")
        loopcls = {"v_mbcnt", "v_lshl_add_u32 (gather address)", "v_mov_b32 .., 0 (zero)", "v_or_b32", "v_dot2_f32_f16", "v_fma_mix", "v_mfma"}
        inner = sum(k for (c, name), k in tot.items() if c == "V" and name in loopcls)
        print(f"\nstep loop (rank, address, zero, or, FMA): {inner} = {inner / a.tiles:.2f} per tile;  everything else: {v - inner} = {(v - inner) / a.tiles:.2f} per tile")
        return
    # basic blocks by label
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        if t:
            blocks[cur].append(t)
    # the block loop: from the first label that is a backward-branch target whose body holds s_barrier
    names = list(blocks)
    pos = {n: i for i, n in enumerate(names)}
    loop = None
    for n in names:
        for ins in blocks[n]:
            m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", ins)
            if m and pos[m.group(1)] <= pos[n]:
                lo, hi = pos[m.group(1)], pos[n]
                if any("s_barrier" in x for k in names[lo:hi + 1] for x in blocks[k]):
                    if loop is None or hi - lo > loop[1] - loop[0]:
                        loop = (lo, hi)
    if loop is None:
        sys.exit("no loop with a barrier found")
    lo, hi = loop
    inloop = names[lo:hi + 1]
    sizes = {n: len(blocks[n]) for n in inloop}
    big = sorted(inloop, key=lambda n: -sizes[n])[:4]          # the four phase bodies: key even, key odd, value even, value odd
    big = sorted(big, key=lambda n: pos[n])
    print(f"kernel {a.kernel}: {sum(len(v) for v in blocks.values())} instructions, block loop = {names[lo]}..{names[hi]} "
          f"({sum(sizes.values())} instructions in the loop's text)")
    print("phase bodies (text order):", ", ".join(f"{n}={sizes[n]}" for n in big if n))
    path = list(inloop) if a.one_body else [n for n in inloop if n not in (big[1], big[3])]
    if a.one_body:
        big = sorted(sorted(inloop, key=lambda n: -sizes[n])[:2], key=lambda n: pos[n])   # key phases (A, B), value phases (A, B)
        big = [big[0], None, big[1], None]
    tot = collections.Counter()
    per_region = collections.OrderedDict()
    for n in path:
        region = "key phase" if n == big[0] else "value phase" if n == big[2] else "around the phases (pointers, softmax step, exchange, barriers)"
        c = per_region.setdefault(region, collections.Counter())
        for ins in blocks[n]:
            c[classify(ins)] += 1
            tot[classify(ins)] += 1

    def show(title, c):
        v = sum(k for (t, _), k in c.items() if t == "V")
        print(f"\n== {title}: {v} vector instructions ({v / a.tiles:.2f} per tile), {sum(c.values()) - v} others")
        for (t, name), k in sorted(c.items(), key=lambda x: (x[0][0] != "V", -x[1])):
            print(f"   {'VALU' if t == 'V' else '    '}  {name:48s} {k:5d}   {k / a.tiles:6.3f} per tile")

    for r, c in per_region.items():
        show(r, c)
    show("one trip of the block loop, one wave (even path)", tot)
    # inside the phases: what is NOT the step loop (mbcnt / address / zero / or / dot2|fma)
    loopcls = {"v_mbcnt", "v_lshl_add_u32 (gather address)", "v_mov_b32 .., 0 (zero)", "v_or_b32", "v_dot2_f32_f16", "v_fma_mix", "v_mfma"}
    inner = sum(k for (t, name), k in tot.items() if t == "V" and name in loopcls)
    v = sum(k for (t, _), k in tot.items() if t == "V")
    print(f"\nstep loop (rank, address, zero, or, FMA): {inner} = {inner / a.tiles:.2f} per tile;  everything else: {v - inner} = {(v - inner) / a.tiles:.2f} per tile")


if __name__ == "__main__":
    main()
