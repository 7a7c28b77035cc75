#!/usr/bin/env python3
"""Static checks of the SpMV kernels' ISA.

(1) No instruction reads an SGPR that a scalar load still has in flight (control-flow aware since round 5: basic blocks, the
pending set propagated forward with a union at joins).

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

(3) The prefetch sink (round 5, pf_at() in spmv.hip): loads whose value is never used are issued inside asm statements into ONE
register per kernel that the compiler does not know is pending; all of a kernel's sink loads must name the same register and no
other instruction of the kernel may write it.

(6) DOT hazard (round 6).  On gfx90a / gfx940 / gfx950 the result of a v_dot* instruction may be read by a DIFFERENT vector instruction only
3 wait states later, and its register be written by one only 4 later (LLVM GCNHazardRecognizer: DotWriteDifferentVALURead /
DotWriteDifferentVALUWrite).  The compiler pads its own instructions; it cannot see a v_dot2 inside an asm statement -- and it does place
moves of the accumulators right behind such a statement (round 6: `v_mov_b32 v29, v9` one wait state behind the key phase's last
`v_dot2_f32_f16 v9`: half the block pairs got a stale score for one head).  So: every asm statement that contains a v_dot* must end with
at least 4 wait states behind its last v_dot* (instructions that do not touch the dot's destination count one each, `s_nop N` counts N + 1),
and inside a statement a non-dot vector instruction may not read / write a dot's destination 3 / 4 wait states or less behind it.

(5) No private segment on the hot launches (round 5).  A kernel with a private segment -- a vector spill, or only the compiler's
emergency slot with no scratch instruction in the text -- is launched with scratch; the key entry point lost 8-10 % to one that a
branch it never takes (the fused decode's window workgroups) had brought in.  The kernels a default call reaches -- every
decode_onepass_sb_kernel, the row kernel, and the entry-point instantiations WITHOUT a window path (template flag WIN = false) -- must
report `.private_segment_fixed_size: 0` and no vector spill.

    python tools/check_smem_hazards.py [--faults-only] [file.s]   # exit code 1 on a hazard; without a file spmv.hip is compiled
                                                                  # (--faults-only: checks (1)-(4), what tools/build_variant.sh gates on)
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


def vgprs(tok):
    """'v[2:5]' -> {2..5}; 'v7' -> {7}; anything else -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _kernels(asm_text):
    """[(kernel name, [(line number, code, in_asm)])]: instructions and .LBB labels of every function, in text order."""
    out, cur, in_asm = [], None, False
    for ln, line in enumerate(asm_text.splitlines(), 1):
        if "#ASMSTART" in line:
            in_asm = True
            if cur is not None:
                cur.append((ln, "#ASMSTART", True))
            continue
        if "#ASMEND" in line:
            in_asm = False
            if cur is not None:
                cur.append((ln, "#ASMEND", True))
            continue
        if "sink_end" in line and cur is not None:
            cur.append((ln, "#SINKEND", True))
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        if code.endswith(":") and not code.startswith("."):
            cur = []
            out.append((code[:-1], cur))
            continue
        if code.startswith(".") and not (code.startswith(".LBB") and code.endswith(":")):
            continue
        if cur is not None:
            cur.append((ln, code, in_asm))
    return out


def _smem_step(code, pending, report):
    """Transfer function of check (1) for one instruction: returns the pending set behind it; report(code) on a hazard."""
    parts = code.replace(",", " ").split()
    op, args = parts[0], parts[1:]
    if op.startswith("s_waitcnt"):
        if "lgkmcnt(0)" in code or re.fullmatch(r"s_waitcnt\s+0", code):
            return set()
        return pending
    if op.startswith("s_load_dword") or op.startswith("s_buffer_load"):
        base = set().union(*[sgprs(x) for x in args[1:]]) if len(args) > 1 else set()
        if base & pending:
            report(code)
        return pending | sgprs(args[0])
    if pending:
        used = set().union(*[sgprs(x) for x in args]) if args else set()
        if used & pending:
            report(code)
    return pending


def check(asm_text):
    """(1) control-flow aware: basic blocks from the .LBB labels and the s_branch / s_cbranch targets, the pending-SGPR set propagated
    forward with a UNION at joins until nothing changes (a wait inside a conditionally skipped block does not clear the set of the path
    around it).  (2) EXEC discipline and (3) the sink register of the prefetch loads walk the text in order (asm statements are
    straight-line code)."""
    hazards = []
    loads = waits = exec_stretches = sinks = 0
    for kernel, ins in _kernels(asm_text):
        # ---- basic blocks
        leaders = {0}
        label_at = {}
        for i, (ln, code, _) in enumerate(ins):
            if code.startswith(".LBB") and code.endswith(":"):
                label_at[code[:-1]] = i
                leaders.add(i)
            op = code.split()[0]
            if (op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64")) and i + 1 < len(ins):
                leaders.add(i + 1)
        starts = sorted(leaders)
        block_of = {}
        for bi, st in enumerate(starts):
            for i in range(st, starts[bi + 1] if bi + 1 < len(starts) else len(ins)):
                block_of[i] = bi
        succ = [[] for _ in starts]
        for bi, st in enumerate(starts):
            en = (starts[bi + 1] if bi + 1 < len(starts) else len(ins)) - 1
            if en < st:
                continue
            code = ins[en][1]
            parts = code.replace(",", " ").split()
            op = parts[0]
            fall = op not in ("s_branch", "s_endpgm", "s_setpc_b64")
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = parts[-1]
                if tgt in label_at:
                    succ[bi].append(block_of[label_at[tgt]])
            if fall and bi + 1 < len(starts):
                succ[bi].append(bi + 1)
        # ---- forward dataflow of the pending set
        inset = [set() for _ in starts]
        outset = [None for _ in starts]
        work = list(range(len(starts)))
        while work:
            bi = work.pop(0)
            pend = set(inset[bi])
            st = starts[bi]
            en = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
            for i in range(st, en):
                code = ins[i][1]
                if code.startswith(".LBB") or code.startswith("#"):
                    continue
                pend = _smem_step(code, pend, lambda c: None)
            if outset[bi] is None or pend != outset[bi]:
                outset[bi] = pend
                for sb in succ[bi]:
                    if not pend <= inset[sb]:
                        inset[sb] |= pend
                        if sb not in work:
                            work.append(sb)
                    elif outset[sb] is None and sb not in work:
                        work.append(sb)
        # ---- report with the converged sets
        for bi, st in enumerate(starts):
            pend = set(inset[bi])
            en = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
            for i in range(st, en):
                ln, code, _ = ins[i]
                if code.startswith(".LBB") or code.startswith("#"):
                    continue
                op = code.split()[0]
                if op.startswith("s_load_dword") or op.startswith("s_buffer_load"):
                    loads += 1
                if op.startswith("s_waitcnt") and ("lgkmcnt(0)" in code or re.fullmatch(r"s_waitcnt\s+0", code)):
                    waits += 1
                pend = _smem_step(code, pend, lambda c, ln=ln: hazards.append((kernel, ln, c)))
        # ---- (2) EXEC discipline, (3) sink register: text order
        in_asm = exec_masked = False
        divergent, saved = 0, []
        until_label = set()   # saved masks whose restore stands at a label ABOVE (a rotated loop): the region is the block that follows, up to the next label
        label_at = {code.split(":")[0]: k for k, (_, code, _) in enumerate(ins) if code.startswith(".LBB")}
        def exits_backward(k):   # behind ins[k]: `s_cbranch_execz <a label already passed>`, or `s_cbranch_execnz <loop>` + `s_branch <a label already passed>`
            nxt = [ins[j][1].split() for j in range(k + 1, min(k + 3, len(ins)))]
            back = lambda w, op: len(w) == 2 and w[0] == op and label_at.get(w[1], len(ins)) < k
            if nxt and back(nxt[0], "s_cbranch_execz"):
                return True
            return len(nxt) == 2 and back(nxt[0], "s_cbranch_execnz") and back(nxt[1], "s_branch")
        sink_regs, sink_lines, vwrites, sink_end = set(), set(), [], None
        for k, (ln, code, _) in enumerate(ins):
            if code == "#SINKEND":
                sink_end = ln
                continue
            if code == "#ASMSTART":
                in_asm = True
                continue
            if code == "#ASMEND":
                if exec_masked:
                    hazards.append((kernel, ln, "asm statement ends with a bitmap in EXEC"))
                in_asm = exec_masked = False
                continue
            if code.startswith(".LBB"):
                if until_label:
                    saved = [r for r in saved if r not in until_label]
                    until_label.clear()
                    divergent = len(saved)
                continue
            parts = code.replace(",", " ").split()
            op, args = parts[0], parts[1:]
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
                if op == "global_load_dword" and len(args) >= 3 and sgprs(args[2]):   # pf_at(): the sink load
                    sink_regs |= vgprs(args[0])
                    sink_lines.add(ln)
                    sinks += 1
            elif "saveexec" in op and args:
                saved.append(args[0])          # (the register pair that holds the mask to come back to)
                if exits_backward(k):          # (round 6: the restore stands at a label above -- the last `if` of a rotated loop body)
                    until_label.add(args[0])
                divergent = len(saved)
            elif op == "s_andn2_b64" and args[:2] == ["exec", "exec"] and len(args) > 2 and args[2] not in saved:
                if not exits_backward(k):      # (a loop whose exit label stands above restores there: nothing below this back edge is inside it)
                    saved.append(args[2])      # (a divergent loop sheds lanes; `s_or_b64 exec, exec, <the same pair>` brings them back)
                divergent = len(saved)
            elif op == "s_or_b64" and args[:2] == ["exec", "exec"] and saved:
                # restoring a mask closes its region AND every region opened inside it (round 6: an inner `if` at the very end of an outer one is
                # closed by the outer restore alone -- counting restores against saves left the depth at 1 for the rest of the kernel)
                if len(args) > 2 and args[2] in saved:
                    del saved[saved.index(args[2]):]
                else:
                    saved.pop()
                divergent = len(saved)
            # destination registers of vector instructions / loads (first operand; stores and compares write no VGPR)
            if args and (op.startswith("v_") or op.startswith("ds_read") or op.startswith("buffer_load") or op.startswith("global_load") or
                         op.startswith("flat_load") or op.startswith("scratch_load")) and not op.startswith("v_cmp") and not op.startswith("v_writelane_b32_dummy"):
                vwrites.append((ln, vgprs(args[0]), code))
        # ---- (4) a vector-memory instruction inside an asm statement that reads an SGPR a VECTOR instruction wrote fewer than five
        # wait states earlier (v_readlane of a spilled pointer in front of pf_at's global_load: the compiler's hazard recogniser
        # does not look inside asm statements)
        hist = []   # (code, in_asm) in text order, the kernel's straight-line view
        for ln, code, ia in ins:
            if code.startswith("#") or code.startswith(".LBB"):
                if code.startswith(".LBB"):
                    hist = []       # (a label: predecessors unknown here; the compiler-made code in front of a join is hazard-free by itself)
                continue
            parts = code.replace(",", " ").split()
            op, args = parts[0], parts[1:]
            if ia and (op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_")):
                need = set().union(*[sgprs(x) for x in args]) if args else set()
                states = 0
                for pcode in reversed(hist):
                    pparts = pcode.replace(",", " ").split()
                    pop, pargs = pparts[0], pparts[1:]
                    if pop.startswith("v_") and pargs and sgprs(pargs[0]) & need:
                        hazards.append((kernel, ln, f"{code}  <- {pcode} ({states} wait states)"))
                        break
                    states += int(pargs[0]) + 1 if pop == "s_nop" and pargs else 1
                    if states >= 5:
                        break
            hist.append(code)
            if len(hist) > 12:
                hist.pop(0)
        if len(sink_regs) > 1:
            hazards.append((kernel, min(sink_lines), f"prefetch sink loads name more than one register: v{sorted(sink_regs)}"))
        # between the first sink load and the end of the sink's life (the `; sink_end` marker of the kernel's last asm statement on it; text
        # order: the prologue and the window workgroups' path, which hold no sink load, lie outside) nothing else may write the register
        for ln, regs, code in vwrites:
            if sink_lines and ln not in sink_lines and regs & sink_regs and min(sink_lines) < ln < (sink_end or 1 << 60):
                hazards.append((kernel, ln, "the prefetch sink register is written by: " + code))
    return hazards, loads, waits, exec_stretches, sinks


_SCRATCH_FREE = (r"decode_onepass_sb_kernel", r"decode_onepass_small_kernel", r"onepass_finish_kernel", r"key_lean_kernelILi\dELi\dELi\dELb0E", r"value_lean_kernelILi\dELi\dELb0E",
                 r"value_spmv_kernelILi\dELb[01]ELi\dELi\dELb0E", r"value_combine_kernel")


def check_dot_hazards(asm_text):
    """Check (6): [(kernel, line, text)] of asm statements whose v_dot* results are not covered by the wait states the hardware needs."""
    bad, n_dots = [], 0
    for kernel, ins in _kernels(asm_text):
        in_asm = False
        recent = []          # [(dest vgprs, wait states since, line, code)] of dots inside the current statement
        for ln, code, _ in ins:
            if code == "#ASMSTART":
                in_asm, recent = True, []
                continue
            if code == "#ASMEND":
                for dest, ws, l0, c0 in recent:
                    if ws < 4:
                        bad.append((kernel, l0, c0 + f"   <- only {ws} wait state(s) to the end of its asm statement"))
                in_asm, recent = False, []
                continue
            if not in_asm or code.startswith(".") or code.startswith("#"):
                continue
            parts = code.replace(",", " ").split()
            op, args = parts[0], parts[1:]
            if op.startswith("v_dot"):
                n_dots += 1
                # (the same dot opcode accumulating into the register -- SrcC -- is fine back to back; as SrcA / SrcB it is not)
                for dest, ws, l0, c0 in recent:
                    if ws < 3 and args[1:3] and (vgprs(args[1]) | vgprs(args[2])) & dest:
                        bad.append((kernel, ln, code + "   <- reads a dot result as a factor too early"))
                recent = [(d, w + 1, l0, c0) for d, w, l0, c0 in recent]
                recent.append((vgprs(args[0]), 0, ln, code))
                continue
            step = 1
            if op == "s_nop":
                step = int(args[0], 0) + 1
            elif op.startswith("v_"):
                rd = set().union(*[vgprs(x) for x in args[1:]]) if len(args) > 1 else set()
                wr = vgprs(args[0]) if args else set()
                for dest, ws, l0, c0 in recent:
                    if (rd & dest and ws < 3) or (wr & dest and ws < 4):
                        bad.append((kernel, ln, code + f"   <- touches the result of `{c0}` {ws} wait state(s) behind it"))
            recent = [(d, w + step, l0, c0) for d, w, l0, c0 in recent]
    return bad, n_dots


def check_private_segments(asm_text):
    """Check (5): [(kernel, bytes of private segment, vector spills)] of the hot kernels that have either."""
    bad, seen = [], 0
    for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", asm_text):
        name, priv, vspill = m.group(1), int(m.group(2)), int(m.group(3))
        if any(re.search(pat, name) for pat in _SCRATCH_FREE):
            seen += 1
            if priv or vspill:
                bad.append((name, priv, vspill))
    return bad, seen


def _report_private(asm_text):
    bad, seen = check_private_segments(asm_text)
    print(f"hot kernels checked for a private segment: {seen}, with one: {len(bad)}")
    for name, priv, vspill in bad:
        print(f"  {name}: private segment {priv} bytes, vector spills {vspill}")
    return 1 if bad or not seen else 0


def main(asm_file=None, faults_only=False):
    """faults_only: checks (1)-(4) -- what can turn into a wrong address or a lane switched on behind the program's back -- without (5), which is
    about speed (tools/build_variant.sh runs this form on every variant's ISA before it emits a library)."""
    if asm_file:
        text = open(asm_file).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "spmv.s")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16", "--cuda-device-only",
                                   "-S", "-I" + os.path.join(ROOT, "include"), SRC, "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
    hazards, loads, waits, stretches, sinks = check(text)
    print(f"scalar loads: {loads}, draining waits: {waits}, EXEC-masked stretches restored: {stretches}, prefetch sink loads: {sinks}, hazards: {len(hazards)}")
    for k, ln, code in hazards[:20]:
        print(f"  {k}: line {ln}: {code}")
    dots, n_dots = check_dot_hazards(text)
    print(f"v_dot* inside asm statements: {n_dots}, uncovered dot results: {len(dots)}")
    for k, ln, code in dots[:20]:
        print(f"  {k}: line {ln}: {code}")
    if hazards or dots:
        return 1
    return 0 if faults_only else _report_private(text)


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if x != "--faults-only"]
    sys.exit(main(args[0] if args else None, faults_only="--faults-only" in sys.argv[1:]))
