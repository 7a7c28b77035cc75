#!/bin/bash
# Build a variant of libmustafar_hip.so with extra hipcc flags (probe / experiment builds; select one at run time with
# MUSTAFAR_HIP_LIB=mustafar_amd/lib/variants/libmustafar_hip_<name>.so).  Usage: tools/build_variant.sh <name> [flags...]
#
# Round 6: NO library is emitted unless the variant's ISA passes tools/check_smem_hazards.py --faults-only (no instruction reads a
# scalar register a load is still writing, EXEC discipline of the asm helpers, the v_readlane -> memory-address wait states).  A build
# that fails those checks can use a half-written register as an ADDRESS: round 5's "no metadata wait" probe took a GPU box down twice
# that way.  The one exception is a knob on the list below -- builds whose in-flight registers are only ever OPERAND VALUES of arithmetic
# (wrong results, no wrong address): they are built with a warning.  Everything else is refused (exit code 3).
#   MUSTAFAR_VARIANT_CHECK_ONLY=1: stop behind the check (exit code 0 / 3), build nothing.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
CANNOT_FAULT="MUSTAFAR_PROBE_NOLDSWAIT MUSTAFAR_DOT_GUARD"   # stale OPERANDS of arithmetic, never an address: the FMAs not waiting for the gathered values /
                                                           # coefficients; the wait states behind a v_dot2 left out (timing A/B of the guard itself)
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function)
TMP="$(mktemp -d)"; trap 'rm -rf "$TMP"' EXIT
hipcc "${FLAGS[@]}" "$@" --cuda-device-only -S -I"$ROOT/include" -o "$TMP/spmv.s" "$ROOT/mustafar_amd/csrc/spmv.hip"
if ! python3 "$ROOT/tools/check_smem_hazards.py" --faults-only "$TMP/spmv.s" > "$TMP/check.txt" 2>&1; then
    allowed=0
    for f in "$@"; do
        k="${f#-D}"; k="${k%%=*}"
        for ok in $CANNOT_FAULT; do [ "$f" != "$k" ] && [ "$k" = "$ok" ] && allowed=1; done
    done
    cat "$TMP/check.txt" >&2
    if [ "$allowed" = 0 ]; then
        echo "build_variant.sh: REFUSED -- variant '$NAME' ($*) fails the ISA checks and none of its knobs is on the cannot-fault list ($CANNOT_FAULT); no library written" >&2
        rm -f "$ROOT/mustafar_amd/lib/variants/libmustafar_hip_$NAME.so"
        exit 3
    fi
    echo "build_variant.sh: warning -- variant '$NAME' fails the ISA checks; built because a knob on the cannot-fault list is set (results are wrong by design)" >&2
fi
[ -n "$MUSTAFAR_VARIANT_CHECK_ONLY" ] && { echo "build_variant.sh: '$NAME' passes the ISA checks (check only, nothing built)"; exit 0; }
mkdir -p "$ROOT/mustafar_amd/lib/variants"
hipcc "${FLAGS[@]}" -fPIC -shared "$@" -o "$ROOT/mustafar_amd/lib/variants/libmustafar_hip_$NAME.so" "$ROOT"/mustafar_amd/csrc/*.hip
echo "build_variant.sh: wrote mustafar_amd/lib/variants/libmustafar_hip_$NAME.so"
