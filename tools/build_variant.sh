#!/bin/bash
# Build a variant of libmustafar_hip.so with extra hipcc flags (probe / experiment builds; select one at run time with
# MUSTAFAR_HIP_LIB=mustafar_amd/lib/variants/libmustafar_hip_<name>.so).  Usage: tools/build_variant.sh <name> [flags...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
mkdir -p "$ROOT/mustafar_amd/lib/variants"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function "$@" \
    -o "$ROOT/mustafar_amd/lib/variants/libmustafar_hip_$NAME.so" "$ROOT"/mustafar_amd/csrc/*.hip
