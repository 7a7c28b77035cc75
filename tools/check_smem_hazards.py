#!/usr/bin/env python3
"""Static checks of the SpMV kernels' ISA.

(1) No instruction reads an SGPR that a scalar load still has in flight.

The inner loop issues `s_load_dword*` inside asm statements and waits later (`s_waitcnt lgkmcnt(0)`); the compiler does not
know those registers are pending, so nothing but our own placement keeps it from copying or spilling them (`s_mov`,
`v_writelane`) between issue and wait -- which would read stale data without any hardware interlock.  This script compiles
spmv.hip to assembly (device only; works without a GPU) and walks every kernel: SGPRs written by a scalar load are
"pending" until the next `s_waitcnt` that drains lgkmcnt; any read of a pending SGPR is reported.

(2) EXEC discipline of the asm helpers.  fma8 / gather8_clean overwrite EXEC with per-tile bitmaps and end with
`s_mov_b64 exec, -1`; they declare no exec clobber and assume a full wave at entry (every workgroup is a multiple of 64
threads and every call site is wave-uniform).  The compiler knows nothing of this, so the ISA is walked: every asm
statement (`;;#ASMSTART` .. `;;#ASMEND`) that loads EXEC with a bitmap must restore it with `s_mov_b64 exec, -1` before it
ends, with no branch or barrier in between; and at the statement's entry EXEC must be the full wave as far as the enclosing
code shows -- the statement may not sit inside a compiler-made divergent region (`s_and_saveexec` .. `s_or_b64 exec`), where
the restoring -1 would switch lanes on that the program had switched off.

    python tools/check_smem_hazards.py        # exit code 1 on a hazard
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mustafar_amd", "csrc", "spmv.hip")


def sgprs(tok):
    """'s[8:23]' -> {8..23}; 's5' -> {5}; anything else -> empty."""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(asm_text):
    hazards, kernel, pending = [], None, set()
    loads = waits = 0
    in_asm = exec_masked = False    # inside an asm statement / inside its bitmap-masked stretch
    divergent = 0                   # depth of compiler-made divergent regions (saveexec .. or exec)
    exec_stretches = 0
    for ln, line in enumerate(asm_text.splitlines(), 1):
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            if exec_masked:
                hazards.append((kernel, ln, "asm statement ends with a bitmap in EXEC"))
            in_asm = exec_masked = False
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        if code.endswith(":") and not code.startswith("."):
            if not code.startswith(".L") and not code.startswith("BB"):
                kernel, pending, divergent = code[:-1], set(), 0
            continue
        if code.startswith("."):
            continue
        parts = code.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        # ---- (2) EXEC discipline
        if in_asm:
            if op == "s_mov_b64" and args and args[0] == "exec":
                if args[1] == "-1":
                    exec_stretches += exec_masked
                    exec_masked = False
                else:
                    if divergent:
                        hazards.append((kernel, ln, "EXEC-masking asm statement inside a divergent region: " + code))
                    exec_masked = True
            elif exec_masked and (op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_barrier")):
                hazards.append((kernel, ln, "EXEC still holds a bitmap at: " + code))
        elif "saveexec" in op:
            divergent += 1
        elif op == "s_or_b64" and args[:2] == ["exec", "exec"] and divergent:
            divergent -= 1
        if op in ("s_branch", "s_endpgm", "s_setpc_b64") and not in_asm:
            # control does not fall through: what follows is reached by jumps only, from places whose own waits were walked
            # where they stand (the compiler's out-of-line blocks load kernel arguments and jump back to the wait)
            pending = set()
            continue
        if op.startswith("s_waitcnt"):
            if "lgkmcnt(0)" in code or re.fullmatch(r"s_waitcnt\s+0", code):
                pending, waits = set(), waits + 1
            continue
        if op.startswith("s_load_dword") or op.startswith("s_buffer_load"):
            base = set().union(*[sgprs(a) for a in args[1:]])
            if base & pending:
                hazards.append((kernel, ln, code))
            pending |= sgprs(args[0])
            loads += 1
            continue
        if pending:
            used = set().union(*[sgprs(a) for a in args]) if args else set()
            if used & pending:
                hazards.append((kernel, ln, code))
    return hazards, loads, waits, exec_stretches


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "spmv.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16", "--cuda-device-only",
                               "-S", "-I" + os.path.join(ROOT, "include"), SRC, "-o", out], stderr=subprocess.DEVNULL)
        hazards, loads, waits, stretches = check(open(out).read())
    print(f"scalar loads: {loads}, draining waits: {waits}, EXEC-masked stretches restored: {stretches}, hazards: {len(hazards)}")
    for k, ln, code in hazards[:20]:
        print(f"  {k}: line {ln}: {code}")
    return 1 if hazards else 0


if __name__ == "__main__":
    sys.exit(main())
