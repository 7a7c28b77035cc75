#!/bin/bash
# Build libmustafar_hip.so for gfx950 (hipcc cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
SRCS=$(ls "$HERE"/*.hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wall -Wno-unused-function "$@" -o "$OUT/libmustafar_hip.so" $SRCS
