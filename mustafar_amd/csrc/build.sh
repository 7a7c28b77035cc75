#!/bin/bash
# Build libmustafar_hip.so for gfx950 (hipcc cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
# -amdgpu-kernarg-preload-count=16: the first 16 dwords of a kernel's arguments arrive in scalar registers with the wave instead of
# through scalar loads at its start (every launch of the step starts a little sooner: c3 +1.5 % tokens/s, c2 +3 %).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
SRCS=$(ls "$HERE"/*.hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function "$@" -o "$OUT/libmustafar_hip.so" $SRCS
