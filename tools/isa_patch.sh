#!/bin/bash
# Build libmustafar_hip from a (hand-patched) device assembly of spmv.hip -- the tool behind round 6's ISA-level bisect of the DOT hazard
# (profiles/r06_probes.txt item 1): dump the ISA, edit single instructions, run the result on the GPU box.
#   tools/isa_patch.sh dump <out.s> [hipcc flags]         device ISA of spmv.hip as the product build compiles it
#   tools/isa_patch.sh build <patched.s> <name>           -> mustafar_amd/lib/variants/libmustafar_hip_<name>.so (MUSTAFAR_HIP_LIB=...)
# The patched ISA goes through the same gate as tools/build_variant.sh: no library unless tools/check_smem_hazards.py --faults-only passes
# (MUSTAFAR_ISA_PATCH_UNCHECKED=1 builds anyway: for experiments whose POINT is a hazard; never for anything that computes an address
# from a register a load may still be writing).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; LLVM=/opt/rocm/lib/llvm/bin
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16)
case "$1" in
dump) OUT="$2"; shift 2; hipcc "${FLAGS[@]}" "$@" --cuda-device-only -S -I"$ROOT/include" -o "$OUT" "$ROOT/mustafar_amd/csrc/spmv.hip" ;;
build)
    S="$2"; NAME="$3"; W="$(mktemp -d)"; trap 'rm -rf "$W"' EXIT
    if [ -z "$MUSTAFAR_ISA_PATCH_UNCHECKED" ] && ! python3 "$ROOT/tools/check_smem_hazards.py" --faults-only "$S" > "$W/check.txt" 2>&1; then
        tail -25 "$W/check.txt" >&2; echo "isa_patch.sh: REFUSED -- the patched ISA fails the hazard checks; no library written" >&2; exit 3
    fi
    $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$S" -o "$W/dev.o"
    $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared "$W/dev.o" -o "$W/dev.hsaco"
    $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input="$W/dev.hsaco" -output="$W/dev.hipfb"
    hipcc "${FLAGS[@]}" -fPIC --cuda-host-only -c "$ROOT/mustafar_amd/csrc/spmv.hip" -Xclang -fcuda-include-gpubinary -Xclang "$W/dev.hipfb" -o "$W/spmv_host.o"
    hipcc "${FLAGS[@]}" -fPIC -c "$ROOT/mustafar_amd/csrc/compress.hip" -o "$W/compress.o"
    mkdir -p "$ROOT/mustafar_amd/lib/variants"
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/mustafar_amd/lib/variants/libmustafar_hip_$NAME.so" "$W/spmv_host.o" "$W/compress.o"
    echo "isa_patch.sh: wrote mustafar_amd/lib/variants/libmustafar_hip_$NAME.so" ;;
*) echo "usage: $0 dump <out.s> [flags] | build <patched.s> <name>"; exit 2 ;;
esac
