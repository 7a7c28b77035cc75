#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of the HIP sources (hipcc -Rpass-analysis=kernel-resource-usage; compiles
device code only, runs without a GPU).  Usage: python tools/kres.py [file.hip ...] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle_short(name: str) -> str:
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name
    out = out.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", out)[:70]


def table(src: str, extra):
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16", "--cuda-device-only", "-c", src,
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "x.o")] + extra
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: +(\w[\w ]*): (.+?) \[-Rpass", line) or re.search(r"remark:\s+(\w[\w ]*): (.+?) \[-Rpass", line)
        if not m:
            m = re.search(r":\d+:\d+: remark: +(\w[\w ]*?): (.+?) \[", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k in ("Function Name", "Name"):
            cur = {"name": demangle_short(v)}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    srcs = args or [os.path.join(ROOT, "mustafar_amd", "csrc", f) for f in ("spmv.hip", "compress.hip")]
    for s in srcs:
        print("==", os.path.relpath(s, ROOT))
        print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spillS':>6s} {'spillV':>6s} {'scr':>5s} {'occ':>4s} {'LDS':>7s}")
        for r in table(s, extra):
            g = lambda k: r.get(k, "?").split()[0]
            print(f"{r['name']:70s} {g('VGPRs'):>5s} {g('AGPRs'):>5s} {g('TotalSGPRs'):>5s} {g('SGPRs Spill'):>6s} {g('VGPRs Spill'):>6s} "
                  f"{g('ScratchSize [bytes/lane]'):>5s} {g('Occupancy [waves/SIMD]'):>4s} {g('LDS Size [bytes/block]'):>7s}")


if __name__ == "__main__":
    main()
